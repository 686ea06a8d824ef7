"""Forward-pass time of the DnCNN plan at B images, conv mode argv[2]; PNP_HIP_LIB selects an A/B build (tools/ab_tree.sh)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_svrg_amd import ops
B = int(sys.argv[1]); mode = int(sys.argv[2])
Wr = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
x = torch.rand(B, 256, 256, device='cuda')
plan = ops.DncnnPlan(Wr, 256, 256, B, winograd=mode)
out = torch.empty_like(x)
for _ in range(2): plan.forward(x, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): plan.forward(x, out)
e1.record(); torch.cuda.synchronize()
ref = ops.DncnnPlan(Wr, 256, 256, B, winograd=5).forward(x)
print(f'mode {mode} B={B}: {e0.elapsed_time(e1) / 5:.3f} ms per forward pass; max |out - mode 5| {float((out - ref).abs().max()):.2e}', flush=True)
