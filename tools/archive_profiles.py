#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of the last tools/prof_bench.sh run from gpurun_out/ (scratch) into profiles/ (tracked).

    python tools/archive_profiles.py r01d_bench_dncnn_wino_B16 "note for the header"

Writes <tag>.json (the bench line of the profiled run), <tag>_kernel_stats.csv (--kernel-trace --stats),
<tag>_pmc_summary.txt (mean FETCH_SIZE / WRITE_SIZE per dispatch and kernel, separate --pmc passes) and updates
profiles/traffic.json (raw FETCH+WRITE bytes per conv launch, the `roofline.traffic` bench.py reports)."""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')
tag = sys.argv[1]
note = sys.argv[2] if len(sys.argv) > 2 else ''


def newest(pattern):
    files = glob.glob(os.path.join(OUT, pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    return re.sub(r'\(.*', '', name)


shutil.copy(os.path.join(OUT, 'bench_prof.json'), os.path.join(ROOT, 'profiles', tag + '.json'))
shutil.copy(newest('prof_stats/**/*kernel_stats.csv'), os.path.join(ROOT, 'profiles', tag + '_kernel_stats.csv'))
lines = ['# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline',
         '# mean RAW counter value per dispatch (KiB); gfx950: double FETCH_SIZE for 16 B/lane streaming reads (MI355X_MICROARCH.md); ' + note]
conv = {}
for ctr, d in (('FETCH_SIZE', 'prof_pmc1'), ('WRITE_SIZE', 'prof_pmc2')):
    agg = {}
    with open(newest(d + '/**/*counter_collection.csv')) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != ctr:
                continue
            k = short(row['Kernel_Name'])
            if not k.startswith(('pnp::', 'void pnp::')):
                continue
            s = agg.setdefault(k, [0, 0.0])
            s[0] += 1
            s[1] += float(row['Counter_Value'])
    for k in sorted(agg):
        n, tot = agg[k]
        lines.append(f'{ctr} {k:<48s} dispatches={n:3d} mean_KiB={tot / n:.1f}')
        if 'k_mid' in k:
            conv[ctr] = tot / n
open(os.path.join(ROOT, 'profiles', tag + '_pmc_summary.txt'), 'w').write('\n'.join(lines) + '\n')
if len(conv) == 2:
    bench = json.load(open(os.path.join(OUT, 'bench_pmc1.json')))
    B = bench['config']['batch_per_gpu']
    tj = os.path.join(ROOT, 'profiles', 'traffic.json')
    t = json.load(open(tj)) if os.path.exists(tj) else {}
    # MI355X_MICROARCH.md, HBM section: the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of
    # wide (16 B/lane) streaming reads -- the conv kernel's LDS-DMA is that case -- so it is doubled; WRITE_SIZE is exact
    t[f'k_mid_B{B}'] = (2.0 * conv['FETCH_SIZE'] + conv['WRITE_SIZE']) * 1024.0
    t['note'] = (f'(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 per conv launch at B={B} (gfx950 correction of the guide: FETCH_SIZE '
                 f'counts 64 B per 128-B request), separate --pmc passes; see profiles/{tag}_pmc_summary.txt')
    json.dump(t, open(tj, 'w'))
print('\n'.join(lines[:3]), '...')
print(open(os.path.join(ROOT, 'profiles', tag + '_kernel_stats.csv')).read()[:600])
