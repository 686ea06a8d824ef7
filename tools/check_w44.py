"""F(4x4,3x3) conv kernel (mode 5) against the direct form and the reference network's golden output; timing vs F(4,3)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnp_svrg_amd import ops
W = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
io = np.load(os.path.join(ROOT, 'tests/golden/dncnn_io.npz')) if os.path.exists(os.path.join(ROOT, 'tests/golden/dncnn_io.npz')) else None
rng = np.random.default_rng(0)
for (H, Wd, B) in ((64, 64, 1), (256, 256, 1), (256, 256, 5), (72, 128, 3)):
    x = torch.from_numpy(rng.random((B, H, Wd)).astype(np.float32)).cuda()
    r0 = ops.DncnnPlan(W, H, Wd, B, winograd=0).forward(x).cpu().numpy()
    r5 = ops.DncnnPlan(W, H, Wd, B, winograd=5).forward(x).cpu().numpy()
    print(f'{H}x{Wd} B={B}: max |F(4x4,3x3) - direct| = {np.abs(r5 - r0).max():.3e}  (max |direct| {np.abs(r0).max():.3f})', flush=True)
if len(sys.argv) > 1:
    Bt = int(sys.argv[1])
    for mode in (4, 5):
        plan = ops.DncnnPlan(W, 256, 256, Bt, winograd=mode)
        x = torch.rand(Bt, 256, 256, device='cuda')
        out = torch.empty_like(x)
        for _ in range(3): plan.forward(x, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n): plan.forward(x, out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f'mode {mode} B={Bt}: {ms:.3f} ms/forward = {ms/15:.3f} ms per mid layer (incl. first/last)', flush=True)
