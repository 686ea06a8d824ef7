# FETCH/WRITE per conv launch and timing for the fp32 Winograd and split-fp16 conv kernels
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for m in f32-winograd f16x3; do
  mkdir -p gpurun_out/pm_${m}_f gpurun_out/pm_${m}_w
  python3 bench.py --conv $m --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['roofline']['launch_ms'])"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pm_${m}_f -- python3 bench.py --conv $m --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pm_${m}_w -- python3 bench.py --conv $m --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv,glob
for ctr,d in (('FETCH_SIZE','gpurun_out/pm_${m}_f'),('WRITE_SIZE','gpurun_out/pm_${m}_w')):
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)[0]
    tot=n=0
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']==ctr and 'k_mid' in r['Kernel_Name']:
            tot+=float(r['Counter_Value']); n+=1
    print('$m',ctr,'per conv launch MiB',tot/n/1024 if n else None,n)
PY
done
