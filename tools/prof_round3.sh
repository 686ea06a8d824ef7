#!/bin/bash
# Round-3 evidence: kernel statistics, HBM traffic (separate --pmc passes) and SQ counters of the config-2 step (one-kernel
# iteration, B = 1024), kernel statistics of the default (config-3) bench, and the default bench line itself.
#     /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/prof_round3.sh'      then      python tools/archive_round3.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03
rm -rf $O && mkdir -p $O
TV="python3 bench.py --workload tv --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tv_stats -- $TV --steps 40 --warmup 10 > $O/tv_stats.json 2> $O/tv_stats.err && echo tv stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/tv_f -- $TV --steps 10 --warmup 0 > $O/tv_f.json 2> $O/tv_f.err && echo tv fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/tv_w -- $TV --steps 10 --warmup 0 > $O/tv_w.json 2> $O/tv_w.err && echo tv write done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/tv_sq1 -- $TV --steps 30 --warmup 10 > $O/tv_sq1.json 2> $O/tv_sq1.err && echo tv sq1 done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $O/tv_sq2 -- $TV --steps 30 --warmup 10 > $O/tv_sq2.json 2> $O/tv_sq2.err && echo tv sq2 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dncnn_stats -- python3 bench.py --no-cpu-baseline --no-secondary > $O/dncnn_stats.json 2> $O/dncnn_stats.err && echo dncnn stats done
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err && echo default bench done
tail -c 600 $O/bench_default.json
