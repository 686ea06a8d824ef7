#!/bin/bash
# memory-pipeline counters (TA / TCP / UTCL1 / TCC / TD) of the conv kernel in bench.py --conv $1 (or, $1 = tv, of k_svrg_outer in --workload tv): one small --pmc pass per group
# (a block collects two to four counters at a time), kernel trace only; progress goes to gpurun_out/mem_$1/log.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CONV=${1:-bf16x3-winograd44}
O=gpurun_out/mem_$CONV
rm -rf $O && mkdir -p $O
B="python3 bench.py --conv $CONV --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
KERNEL=k_mid_wino44
if [ "$CONV" = tv ]; then B="python3 bench.py --workload tv --steps 20 --warmup 10 --no-cpu-baseline --no-secondary"; KERNEL=k_svrg_outer; fi
i=0
for grp in "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" "TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_TOTAL_CYCLES" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY" \
           "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCC_HIT TCC_MISS" "TD_TD_BUSY TD_TC_STALL" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -- $B > $O/p$i.json 2> $O/p$i.err; echo "pass $i ($grp): exit $?" >> $O/log.txt
done
python3 - "$O" "$KERNEL" <<'PY' >> $O/log.txt
import csv, glob, collections, sys
O, KERNEL = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(f'{O}/p*/')):
    fs = glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)
    if not fs: continue
    v = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if KERNEL in r['Kernel_Name']:
            v[r['Counter_Name']].append(float(r['Counter_Value']))
    for k in sorted(v): print(f'{k:36s} {sum(v[k]) / len(v[k]):14.4g}  ({len(v[k])} launches)')
PY
cat $O/log.txt
