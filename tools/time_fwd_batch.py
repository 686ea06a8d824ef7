"""DnCNN-17 forward pass: microseconds per image against the batch size (does a MALL-resident working set help?)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnp_svrg_amd import ops
W = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
for B in (1, 2, 3, 5, 7, 9, 15, 24, 120):
    plan = ops.DncnnPlan(W, 256, 256, B, winograd=5)
    x = torch.rand(B, 256, 256, device='cuda'); out = torch.empty_like(x)
    for _ in range(3): plan.forward(x, out)
    torch.cuda.synchronize()
    n = max(3, 240 // B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): plan.forward(x, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'B={B:4d}: {ms:8.3f} ms/forward = {ms / B * 1e3:7.1f} us/image  (activations 2 x {B * 16.8:.0f} MB)', flush=True)
    del plan
