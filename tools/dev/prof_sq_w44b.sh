#!/bin/bash
# SQ counters of the conv kernel with conv mode 6 (bench --conv bf16x3-winograd44), two --pmc passes, kernel trace only
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/sq6
rm -rf $O && mkdir -p $O
B="python3 bench.py --conv bf16x3-winograd44 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- $B > $O/p1.json 2> $O/p1.err && echo p1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/p2 -- $B > $O/p2.json 2> $O/p2.err && echo p2
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p3 -- $B > $O/p3.json 2> $O/p3.err && echo p3
python3 - <<'PY'
import csv, glob, collections
for d in ('p1', 'p2', 'p3'):
    fs = glob.glob(f'gpurun_out/sq6/{d}/**/*counter_collection.csv', recursive=True)
    if not fs: print(d, 'no file'); continue
    v = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'k_mid_wino44b' in r['Kernel_Name']:
            v[r['Counter_Name']].append(float(r['Counter_Value']))
    for k in sorted(v): print(f'{k:28s} {sum(v[k]) / len(v[k]):14.4g}  ({len(v[k])} launches)')
PY
