#!/usr/bin/env python3
"""Copy the summaries of the last tools/prof_round3.sh run from gpurun_out/r03 (scratch) into profiles/ (tracked):
r03_tv_kernel_stats.csv, r03_tv_pmc_summary.txt, r03_tv_sq_counters.txt, r03_tv.json, r03_dncnn_kernel_stats.csv, r03_dncnn.json,
r03_bench_default.json; updates profiles/traffic.json (`roofline.traffic` of the bench line) with the measured config-2 bytes."""
import collections, csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, 'gpurun_out', 'r03')
P = os.path.join(ROOT, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'


def newest(pattern):
    files = glob.glob(os.path.join(O, pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    return re.sub(r'\(.*', '', name).replace('void ', '')


for src, dst in (('tv_stats.json', f'{tag}_tv.json'), ('dncnn_stats.json', f'{tag}_dncnn.json'), ('bench_default.json', f'{tag}_bench_default.json')):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
for d, dst in (('tv_stats', f'{tag}_tv_kernel_stats.csv'), ('dncnn_stats', f'{tag}_dncnn_kernel_stats.csv')):
    f = newest(d + '/**/*kernel_stats.csv')
    if f:
        shutil.copy(f, os.path.join(P, dst))
# HBM traffic of a config-2 step: all kernels of the profiled steps, per problem-iteration
bench = json.load(open(os.path.join(O, 'tv_f.json')))
B, steps = bench['config']['batch_per_gpu'], bench['steps']
lines = [f'# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --workload tv --steps {steps} --warmup 0 (B = {B}; the warm-up is rounded up to one outer iteration, so two outer iterations are counted)',
         '# counters are in KiB; gfx950: FETCH_SIZE is doubled for 16-byte-per-lane reads (MI355X_MICROARCH.md, HBM section) -- the one-kernel iteration now',
         '# reads and writes 16 bytes per lane everywhere, the draw kernel 4 bytes per lane']
tot = {}
nouter = 0
for ctr, d in (('FETCH_SIZE', 'tv_f'), ('WRITE_SIZE', 'tv_w')):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(newest(d + '/**/*counter_collection.csv'))):
        k = short(r['Kernel_Name'])
        if r['Counter_Name'] == ctr and k.startswith('pnp::'):
            agg[k][0] += 1
            agg[k][1] += float(r['Counter_Value'])
    tot[ctr] = 0.0
    for k, (n, t) in sorted(agg.items()):
        wide = 'k_svrg_iter' in k or 'k_svrg_outer' in k
        corr = t * (2.0 if (ctr == 'FETCH_SIZE' and wide) else 1.0)
        lines.append(f'{ctr} {k:<40s} dispatches={n:4d} mean_KiB_raw={t / n:12.1f}' + ('  (x2 for 16-byte reads)' if ctr == 'FETCH_SIZE' and wide else ''))
        if any(s in k for s in ('k_svrg_iter', 'k_svrg_outer', 'k_draw_thr')):
            tot[ctr] += corr
        if 'k_svrg_outer' in k:
            nouter = n
if nouter:                                    # one k_svrg_outer launch = T2 = 10 inner iterations (bench.py rounds the warm-up up to whole outer iterations)
    steps = 10 * nouter
per = (tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024 / (B * steps)
lines.append(f'config-2 step (k_svrg_outer = folded outer refresh + T2 inner iterations per launch, + k_draw_thr): {per / 1e6:.3f} MB per problem-iteration '
             f'(fetch {tot["FETCH_SIZE"] * 1024 / (B * steps) / 1e6:.3f} + write {tot["WRITE_SIZE"] * 1024 / (B * steps) / 1e6:.3f}); algorithmic 2.425 MB (SURVEY 8d), physically needed 1.57 MB')
open(os.path.join(P, f'{tag}_tv_pmc_summary.txt'), 'w').write('\n'.join(lines) + '\n')
tj = os.path.join(P, 'traffic.json')
t = json.load(open(tj)) if os.path.exists(tj) else {}
t[f'tv_step_B{B}'] = per * B
t[f'tv_step_B{B}_source'] = f'profiles/{tag}_tv_pmc_summary.txt (builder PMC run, not measured by the run that prints this line)'
json.dump(t, open(tj, 'w'), indent=1)
# SQ counters of k_svrg_outer
vals = collections.defaultdict(list)
for d in ('tv_sq1', 'tv_sq2'):
    f = newest(d + '/**/*counter_collection.csv')
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if 'k_svrg_outer' in r['Kernel_Name']:
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
if vals:
    m = {k: sum(v) / len(v) for k, v in vals.items()}
    out = ['# rocprofv3 --kernel-trace --pmc <SQ counters> (two passes; tools/prof_round3.sh) -- bench.py --workload tv --steps 30 --warmup 10, kernel k_svrg_outer (T2 = 10 inner iterations per launch)',
           '# mean per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles (guide)']
    for k in sorted(m):
        out.append(f'{k:30s} {m[k]:14.4g}   ({len(vals[k])} launches)')
    if 'SQ_WAVE_CYCLES' in m:
        wc = m['SQ_WAVE_CYCLES']
        out.append('')
        out.append(f'wave time: instructions issuing {m.get("SQ_ACTIVE_INST_ANY", 0) / wc:.2f} (vector ALU {m.get("SQ_ACTIVE_INST_VALU", 0) / wc:.2f}), '
                   f'waitcnt / barrier {m.get("SQ_WAIT_ANY", 0) / wc:.2f}')
    if 'SQ_WAVES' in m:
        w = m['SQ_WAVES']
        out.append('instructions per wave: ' + ', '.join(f'{n} {m[c] / w:.0f}' for n, c in (('vector ALU', 'SQ_INSTS_VALU'), ('LDS', 'SQ_INSTS_LDS'), ('scalar', 'SQ_INSTS_SALU'),
                                                                                          ('vector-memory reads', 'SQ_INSTS_VMEM_RD'), ('writes', 'SQ_INSTS_VMEM_WR')) if c in m))
    if 'SQ_LDS_BANK_CONFLICT' in m and m.get('SQ_LDS_IDX_ACTIVE'):
        out.append(f'LDS bank-conflict cycles / LDS active cycles = {m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]:.3f}')
    open(os.path.join(P, f'{tag}_tv_sq_counters.txt'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(lines[-4:]))
