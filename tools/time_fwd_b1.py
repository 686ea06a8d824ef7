"""DnCNN-17 forward passes on one 256 x 256 image (for a rocprofv3 --kernel-trace --stats run: tools/prof_b1_dncnn.sh)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnp_svrg_amd import ops
W = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
plan = ops.DncnnPlan(W, 256, 256, 1, winograd=5)
x = torch.rand(1, 256, 256, device='cuda'); out = torch.empty_like(x)
for _ in range(3): plan.forward(x, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): plan.forward(x, out)
e1.record(); torch.cuda.synchronize()
print(f'{e0.elapsed_time(e1) / 200 * 1e3:.1f} us per forward pass')
