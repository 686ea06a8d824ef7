#!/bin/bash
# SQ counters of the conv kernel in the headline bench (one --pmc pass, kernel trace only): issue / wait split, LDS bank conflicts, MFMA busy.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/sq
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/p1.json 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/p2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/p2.json 2> $O/p2.err
echo done
