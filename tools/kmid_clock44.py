"""In-kernel clock and phase shares of the F(4x4,3x3) conv kernel under load (diagnostic build, PNP_DEBUG_STAMPS=1)."""
import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnp_svrg_amd import ops, _native as N
W = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
plan = ops.DncnnPlan(W, 256, 256, B, winograd=5)
x = torch.rand(B, 256, 256, device='cuda'); plan.forward(x)
c, r = ctypes.c_double(), ctypes.c_double()
N.call('pnp_dncnn_debug_clock', plan._h, 300, ctypes.byref(c), ctypes.byref(r), None)
tiles = B * 128 / 256
print(f'cycles={c.value:.0f} ref_ticks={r.value:.0f} -> clock {c.value/r.value*0.1:.3f} GHz; loop {r.value*10/1000:.1f} us; tiles per WG {tiles:.0f}; cycles/tile {c.value/tiles:.0f} (MFMA floor 36864)')
