# inner-iterations/s vs batch size (B in {1, 12, 120, 1024}); prints value, ms/step, roofline frac
for B in 1 12 120 1024; do
  S=20; [ $B -ge 120 ] && S=6; [ $B -ge 1024 ] && S=3
  python bench.py --batch $B --steps $S --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dncnn B=$B', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'])"
done
for B in 1 12 120 1024; do
  python bench.py --workload tv --batch $B --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tv B=$B', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
