#!/bin/bash
# rocprofv3 kernel stats of the config-4 bench (Deblur + NLM + SAGA, B = 64); see tools/prof_round2.sh.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/saga_stats -- python3 bench.py --workload saga-nlm --steps 20 --warmup 3 --no-cpu-baseline > $O/saga_stats.json 2> $O/saga_stats.err
echo "saga done"
