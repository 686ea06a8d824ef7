#!/bin/bash
# Round-2 profiles on the GPU box: rocprofv3 kernel stats + FETCH/WRITE PMC passes for the three bench workloads.
# Outputs under gpurun_out/r02/ ; tools/archive_round2.py copies the summaries into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02
rm -rf $O && mkdir -p $O
# headline (config 3, B = 120)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dncnn_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/dncnn_stats.json 2> $O/dncnn_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/dncnn_f -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/dncnn_f.json 2> $O/dncnn_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/dncnn_w -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $O/dncnn_w.json 2> $O/dncnn_w.err
echo "dncnn done"
# config 2 (TV prox, B = 256)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tv_stats -- python3 bench.py --workload tv --steps 200 --warmup 10 --no-cpu-baseline > $O/tv_stats.json 2> $O/tv_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tv_f -- python3 bench.py --workload tv --steps 20 --warmup 0 --no-cpu-baseline > $O/tv_f.json 2> $O/tv_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tv_w -- python3 bench.py --workload tv --steps 20 --warmup 0 --no-cpu-baseline > $O/tv_w.json 2> $O/tv_w.err
echo "tv done"
# config 4 (Deblur + NLM + SAGA, B = 64)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/saga_stats -- python3 bench.py --workload saga-nlm --steps 20 --warmup 3 --no-cpu-baseline > $O/saga_stats.json 2> $O/saga_stats.err
echo "saga done"
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
