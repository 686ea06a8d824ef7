set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_stats $R/gpurun_out/prof_pmc1 $R/gpurun_out/prof_pmc2
cd $R
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_pmc1 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pmc1.json 2> gpurun_out/bench_pmc1.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_pmc2 -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pmc2.json 2> gpurun_out/bench_pmc2.err
find gpurun_out -name "*.csv" | head -30
