#!/bin/bash
# SQ counters (two --pmc passes, kernel trace only) of a secondary workload's dominant kernel: tools/prof_sq2.sh tv | saga-nlm
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
W=$1
O=gpurun_out/sq_$W
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/p1.json 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $O/p2 -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/p2.json 2> $O/p2.err
echo done $W
