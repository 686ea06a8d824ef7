"""The 3 x bf16 split F(4x4,3x3) conv layer (mode 6) against a float64 convolution and the fp32 kernels; timing vs mode 5.
    python tools/dev/check_w44b.py [B]        (B > 0: also time the 17-layer forward pass at 256 x 256)"""
import os, sys, ctypes, numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pnp_svrg_amd import ops, _native as N
rng = np.random.default_rng(3)
w = {'n_layers': np.int64(3),
     'conv0.weight': rng.standard_normal((64, 1, 3, 3)).astype(np.float32),
     'conv1.weight': (rng.standard_normal((64, 64, 3, 3)) / 24.0).astype(np.float32),
     'conv1.bias': (rng.standard_normal(64) * 0.1).astype(np.float32),
     'conv2.weight': (rng.standard_normal((1, 64, 3, 3)) / 24.0).astype(np.float32)}
for (H, Wd, B) in ((8, 64, 1), (64, 64, 1), (72, 128, 3), (256, 256, 2)):
    x = rng.standard_normal((B, 64, H, Wd)).astype(np.float32)
    ref = F.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w['conv1.weight']).double(),
                          torch.from_numpy(w['conv1.bias']).double(), padding=1)).numpy()
    xd = torch.from_numpy(x).cuda()
    msg = []
    for mode in (0, 5, 6):
        plan = ops.DncnnPlan(w, H, Wd, B, winograd=mode)
        out = torch.full_like(xd, float('nan'))
        plan.debug_mid_layer(0, xd, out)
        torch.cuda.synchronize()
        r = out.cpu().numpy().astype(np.float64)
        e = np.abs(r - ref)
        msg.append(f'mode {mode}: max err {np.nanmax(e):.3e} (rel {np.nanmax(e) / np.abs(ref).max():.2e}) nan {int(np.isnan(r).sum())}')
        if mode == 6 and (np.isnan(r).any() or np.nanmax(e) > 1e-3):
            bad = np.argwhere(~(e < 1e-3))
            print('   first bad (b, c, y, x):', bad[:8].tolist(), ' count', len(bad), 'of', e.size)
            cs = sorted(set(bad[:, 1].tolist())); ys = sorted(set((bad[:, 2] % 8).tolist())); xs = sorted(set((bad[:, 3] % 64).tolist()))
            print('   bad channels', cs[:40], '\n   bad rows mod 8', ys, '\n   bad cols mod 64', xs[:70])
    print(f'{H}x{Wd} B={B}: ' + ' | '.join(msg), flush=True)
if len(sys.argv) > 1 and int(sys.argv[1]) > 0:
    Bt = int(sys.argv[1])
    Wr = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
    x = torch.rand(Bt, 256, 256, device='cuda')
    res = {}
    for mode in (5, 6):
        plan = ops.DncnnPlan(Wr, 256, 256, Bt, winograd=mode)
        out = torch.empty_like(x)
        for _ in range(2): plan.forward(x, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): plan.forward(x, out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        res[mode] = out.cpu().numpy()
        c, r = ctypes.c_double(), ctypes.c_double()
        N.call('pnp_dncnn_debug_clock', plan._h, 30, ctypes.byref(c), ctypes.byref(r), None)
        tiles = Bt * 128 / 256
        print(f'mode {mode} B={Bt}: {ms:.3f} ms per forward pass; clock {c.value / r.value * 0.1:.3f} GHz, cycles per region {c.value / tiles:.0f}', flush=True)
    print(f'max |mode 6 - mode 5| through the 17-layer net: {np.abs(res[6] - res[5]).max():.3e} (max |out| {np.abs(res[5]).max():.3f})')
