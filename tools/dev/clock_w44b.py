"""In-kernel clock of the conv kernel (mode from argv[2], default 6) under load; PNP_W44_VAR selects an ablation build."""
import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['PNP_DEBUG_STAMPS'] = '1'
from pnp_svrg_amd import ops, _native as N
W = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 6
plan = ops.DncnnPlan(W, 256, 256, B, winograd=mode)
x = torch.rand(B, 256, 256, device='cuda'); plan.forward(x)
c, r = ctypes.c_double(), ctypes.c_double()
N.call('pnp_dncnn_debug_clock', plan._h, 30, ctypes.byref(c), ctypes.byref(r), None)
tiles = B * 128 / 256
print(f'VAR={os.environ.get("PNP_W44_VAR", "0")} mode {mode}: clock {c.value/r.value*0.1:.3f} GHz; cycles/region {c.value/tiles:.0f}', flush=True)
