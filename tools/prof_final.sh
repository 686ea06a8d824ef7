#!/bin/bash
# rocprofv3 kernel stats of the headline bench + the default bench line (no profiler) -> gpurun_out/r02; archived by hand under profiles/<tag>_*
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dncnn_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/dncnn_stats.json 2> $O/dncnn_stats.err
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "final done"
