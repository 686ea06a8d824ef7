import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from pnp_svrg_amd import ops
W = dict(np.load('/root/repo/tests/golden/dncnn_noise15.npz'))
import os
mode = int(os.environ.get('WINO', '1'))
for B in (1, 4, 16):
    plan = ops.DncnnPlan(W, 256, 256, B, winograd=mode)
    x = torch.rand(B, 256, 256, device='cuda')
    out = torch.empty_like(x)
    for _ in range(3): plan.forward(x, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): plan.forward(x, out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 72628568064 * B
    print(f'B={B}: {ms:.3f} ms/forward  {fl/ms/1e9:.1f} TFLOP/s  ({fl/ms/1e9/157.3*100:.1f}% of f32 MFMA peak)')
