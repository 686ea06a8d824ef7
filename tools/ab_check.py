"""Correctness of an A/B library build (PNP_HIP_LIB): F(4x4,3x3) conv against the direct kernel on a few shapes."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnp_svrg_amd import ops
W = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
rng = np.random.default_rng(0)
for (H, Wd, B) in ((64, 64, 1), (256, 256, 1), (256, 256, 5), (72, 128, 3)):
    x = torch.from_numpy(rng.random((B, H, Wd)).astype(np.float32)).cuda()
    r0 = ops.DncnnPlan(W, H, Wd, B, winograd=0).forward(x).cpu().numpy()
    r5 = ops.DncnnPlan(W, H, Wd, B, winograd=5).forward(x).cpu().numpy()
    print(os.path.basename(os.environ.get('PNP_HIP_LIB', 'default')), f'{H}x{Wd} B={B}: max |w44 - direct| = {np.abs(r5 - r0).max():.3e}', flush=True)
