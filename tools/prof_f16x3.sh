set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/prof_f16; mkdir -p gpurun_out/prof_f16
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f16 -- python3 bench.py --conv f16x3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_f16_prof.json 2> gpurun_out/bench_f16_prof.err
cat gpurun_out/bench_f16_prof.json | cut -c1-300
