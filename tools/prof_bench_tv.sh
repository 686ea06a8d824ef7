set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_tv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tv -- python3 bench.py --workload tv --batch 256 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/bench_tv.json 2> gpurun_out/bench_tv.err
cat gpurun_out/bench_tv.json
cat gpurun_out/prof_tv/*/*kernel_stats.csv | cut -c1-150
