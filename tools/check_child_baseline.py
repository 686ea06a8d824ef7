"""bench.py's cpu_baseline child (DnCNN workload) run standalone, under a parent that has initialised the GPU, and after GPU work: the
figure moves between 6 and 31 inner-iterations/s with the placement of its 16 threads on the pool's shared hosts."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--cpu-baseline-child', 'dncnn:5']
print('standalone child   :', subprocess.run(cmd, capture_output=True, text=True).stdout.strip()[:60], flush=True)
import torch
torch.zeros(1, device='cuda'); torch.cuda.synchronize()
print('parent has GPU init:', subprocess.run(cmd, capture_output=True, text=True).stdout.strip()[:60], flush=True)
x = torch.rand(8192, 8192, device='cuda')
for _ in range(20): y = x @ x
torch.cuda.synchronize()
print('parent after GPU work:', subprocess.run(cmd, capture_output=True, text=True).stdout.strip()[:60], flush=True)
env = dict(os.environ, OMP_NUM_THREADS='16')
print('child with OMP_NUM_THREADS=16:', subprocess.run(cmd, capture_output=True, text=True, env=env).stdout.strip()[:60], flush=True)
