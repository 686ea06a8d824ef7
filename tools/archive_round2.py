#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of the last tools/prof_round2.sh run (gpurun_out/r02, scratch) into profiles/ (tracked).

    python tools/archive_round2.py r02a

Writes per workload <tag>_<wl>.json (the bench line of the profiled run), <tag>_<wl>_kernel_stats.csv
(--kernel-trace --stats), <tag>_<wl>_pmc_summary.txt (FETCH_SIZE / WRITE_SIZE per kernel, separate --pmc passes) and updates
profiles/traffic.json (what bench.py reports as roofline.traffic).
Counter handling as MI355X_MICROARCH.md (HBM) prescribes: separate passes, values in KiB, FETCH_SIZE doubled on gfx950 (it
tallies 64 B per 128-B request of a wide streaming read; the 4-byte-per-lane loads of the FFT kernels are 'uncalibrated'
there -- doubled all the same, which can only over-state their traffic)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'r02')
PROF = os.path.join(ROOT, 'profiles')
tag = sys.argv[1]


def one(pattern):
    files = glob.glob(os.path.join(OUT, pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    return re.sub(r'\(.*', '', name).replace('void ', '')


def counters(wl):
    agg = collections.defaultdict(lambda: {'n': 0, 'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})
    for ctr, d in (('FETCH_SIZE', wl + '_f'), ('WRITE_SIZE', wl + '_w')):
        f = one(d + '/**/*counter_collection.csv')
        if f is None:
            return None
        n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = short(row['Kernel_Name'])
            if row['Counter_Name'] != ctr or 'pnp::' not in k:
                continue
            agg[k][ctr] += float(row['Counter_Value'])
            n[k] += 1
        for k, v in n.items():
            agg[k]['n'] = v
    return agg


tj = os.path.join(PROF, 'traffic.json')
traffic = json.load(open(tj)) if os.path.exists(tj) else {}
for wl in ('dncnn', 'tv', 'saga'):
    st = one(wl + '_stats/**/*kernel_stats.csv')
    if st is None:
        continue
    shutil.copy(st, os.path.join(PROF, f'{tag}_{wl}_kernel_stats.csv'))
    shutil.copy(os.path.join(OUT, wl + '_stats.json'), os.path.join(PROF, f'{tag}_{wl}.json'))
    agg = counters(wl)
    if agg is None:
        continue
    bench = json.loads(open(os.path.join(OUT, wl + '_f.json')).read().strip().splitlines()[-1])
    B, steps = bench['config']['batch_per_gpu'], bench['steps']
    lines = [f'# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- bench.py --workload {wl} (B = {B}, {steps} timed steps)',
             '# per kernel: dispatches, mean RAW counter per dispatch (KiB), corrected bytes per dispatch = (2 x FETCH + WRITE) KiB x 1024']
    step_total = 0.0
    for k in sorted(agg):
        a = agg[k]
        f, w = a['FETCH_SIZE'] / a['n'], a['WRITE_SIZE'] / a['n']
        byt = (2 * f + w) * 1024
        lines.append(f'{k:<44s} dispatches={a["n"]:4d} FETCH_SIZE={f:10.1f} WRITE_SIZE={w:10.1f} corrected={byt / 1e6:9.2f} MB')
        if wl == 'dncnn' and 'k_mid' in k:
            traffic[f'k_mid_B{B}'] = byt
        if wl == 'tv' and any(t in k for t in ('k_svrg_iter', 'k_draw_thr', 'k_rows_fwd', 'k_cols', 'k_rows_inv', 'k_prox_tv')):
            step_total += byt * a['n']
    if wl == 'tv':
        per_step = step_total / steps
        traffic[f'tv_step_B{B}'] = per_step
        traffic['tv_note'] = (f'sum over the kernels of the timed config-2 steps at B={B} (one-kernel iterations, 1/T2 of the draw and '
                              f'full-gradient launches) of (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024, per step; profiles/{tag}_tv_pmc_summary.txt')
        lines.append(f'whole step: {per_step / 1e6:.1f} MB for {B} problems = {per_step / B / 1e6:.3f} MB per problem-iteration (algorithmic 2.425 MB)')
    if wl == 'dncnn':
        traffic['note'] = (f'(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 per conv launch (gfx950 correction of the guide), separate --pmc '
                           f'passes; profiles/{tag}_dncnn_pmc_summary.txt')
    open(os.path.join(PROF, f'{tag}_{wl}_pmc_summary.txt'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
json.dump(traffic, open(tj, 'w'))
for wl in ('dncnn', 'tv', 'saga'):
    f = os.path.join(PROF, f'{tag}_{wl}_kernel_stats.csv')
    if os.path.exists(f):
        print(wl)
        for r in list(csv.DictReader(open(f)))[:7]:
            print('  ', short(r['Name'])[:50].ljust(50), r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us', r['Percentage'])
