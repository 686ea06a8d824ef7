#!/bin/bash
# Throughput vs batch size (SURVEY 8d list) for configs 3 and 2 on one MI355X -> gpurun_out/batch_sweep.txt
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && : > gpurun_out/batch_sweep.txt
for wl in dncnn tv; do
  for b in 1 12 120 1024; do
    steps=20; [ $wl = tv ] && steps=200; [ $b = 1024 ] && [ $wl = dncnn ] && steps=5
    timeout -k 10 300 python bench.py --workload $wl --batch $b --steps $steps --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | \
      python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('$wl', l['config']['batch_per_gpu'], l['value'], l['ms_per_step'], r['frac'], r.get('algorithmic_tflops',''))" >> gpurun_out/batch_sweep.txt
  done
done
cat gpurun_out/batch_sweep.txt
