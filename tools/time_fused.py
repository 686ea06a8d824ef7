"""Phase-by-phase timing of the one-kernel iteration (PNP_FUSED_STOP = k leaves after phase k)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnp_svrg_amd.engine import CsmriBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = CsmriBatch.synthetic(B, 256, 256, 0.2, 20.0, seed=1)
p = b.plan
z, w, mu = b.xinit.clone(), b.xinit.clone() * 0.9, b.xinit.clone() * 1e-4
sel = torch.empty((1, B, 256, 8), dtype=torch.int32, device='cuda')
p.draw_thresholds(b.bits, 1000, 1, 0, 1, selbits=sel)
sse = torch.empty(B, dtype=torch.float64, device='cuda')
out = torch.empty_like(z)
for stop in (10, 1, 2, 3, 4, 0):
    os.environ['PNP_FUSED_STOP'] = str(stop)
    for _ in range(3):
        p.svrg_step(z, w, sel[0], alpha=-2.0, beta=1.0, c1=z, gamma=-2e3, c2=mu, out=out, xrec=b.xrec, sse=sse)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        p.svrg_step(z, w, sel[0], alpha=-2.0, beta=1.0, c1=z, gamma=-2e3, c2=mu, out=out, xrec=b.xrec, sse=sse)
    e1.record(); torch.cuda.synchronize()
    print(f'stop after phase {"1 (loads only)" if stop == 10 else (stop or 5)}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per launch (B = {B})')
