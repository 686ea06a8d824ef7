#!/bin/bash
# rocprofv3 kernel stats + FETCH/WRITE PMC passes of the headline bench (config 3, B = 120) only; see tools/prof_round2.sh.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dncnn_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/dncnn_stats.json 2> $O/dncnn_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/dncnn_f -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/dncnn_f.json 2> $O/dncnn_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/dncnn_w -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/dncnn_w.json 2> $O/dncnn_w.err
echo "dncnn done"
