import sys, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo')
from pnp_svrg_amd import ops, _native as N
W = dict(np.load('/root/repo/tests/golden/dncnn_noise15.npz'))
for mode in (3, 1):
    plan = ops.DncnnPlan(W, 256, 256, 16, winograd=mode)
    x = torch.rand(16, 256, 256, device='cuda'); plan.forward(x)
    c, r = ctypes.c_double(), ctypes.c_double()
    N.call('pnp_dncnn_debug_clock', plan._h, 400, ctypes.byref(c), ctypes.byref(r), None)
    print(f'mode {mode}: cycles={c.value:.0f} ref_ticks={r.value:.0f} -> clock {c.value/r.value*0.1:.3f} GHz; {r.value*10/1000:.1f} us per launch; cycles/tile {c.value/16:.0f}')
