import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnp_svrg_amd import ops
W = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
plan = ops.DncnnPlan(W, 256, 256, 120, winograd=5)
x = torch.rand(120, 256, 256, device='cuda'); out = torch.empty_like(x)
for _ in range(6): plan.forward(x, out)
torch.cuda.synchronize()
