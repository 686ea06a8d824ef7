# HBM-side traffic of one config-2 inner iteration (all kernels of a step), B = 256: separate --pmc passes
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/tv_f gpurun_out/tv_w; mkdir -p gpurun_out/tv_f gpurun_out/tv_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/tv_f -- python3 bench.py --workload tv --batch 256 --steps 20 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/tv_w -- python3 bench.py --workload tv --batch 256 --steps 20 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for ctr, d in (('FETCH_SIZE', 'gpurun_out/tv_f'), ('WRITE_SIZE', 'gpurun_out/tv_w')):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if r['Counter_Name'] == ctr and 'pnp::' in k and any(t in k for t in ('k_rows_fwd', 'k_cols', 'k_rows_inv', 'k_prox_tv', 'k_draw_mb')):
            agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
    for k, (n, tot) in sorted(agg.items()):
        print(ctr, k, 'dispatches', n, 'mean KiB', round(tot / n, 1))
        out.setdefault(k, {})[ctr] = tot / n
json.dump(out, open('gpurun_out/tv_traffic.json', 'w'))
PY
