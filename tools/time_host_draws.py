"""Host-side cost of the legacy minibatch draw (np.random.choice vs pnp_legacy_choice), before and after the GPU is initialised."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pnp_svrg_amd import legacy_rng
print('cpus available:', len(os.sched_getaffinity(0)), flush=True)
pool = np.sort(np.random.RandomState(0).choice(65536, 13107, replace=False))
def bench(tag):
    for name, f in (('np.random.choice', lambda: np.random.choice(pool, 1000, replace=False)), ('pnp_legacy_choice', lambda: legacy_rng.choice(pool, 1000))):
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(100): f()
            ts.append((time.perf_counter() - t0) / 100 * 1e6)
        print(f'{tag:28s} {name:18s} us per draw: ' + ' '.join('%.0f' % t for t in ts), flush=True)
bench('before GPU init')
import torch
torch.zeros(1, device='cuda'); torch.cuda.synchronize()
bench('after GPU init')
x = torch.rand(64, 1024, 1024, device='cuda')
for _ in range(50): x = x * 1.0001
bench('GPU busy (queued kernels)')
torch.cuda.synchronize()
bench('after synchronize')
