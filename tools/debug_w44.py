import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnp_svrg_amd import ops
rng = np.random.default_rng(1)
H = W = 64
x = torch.from_numpy(rng.random((1, H, W)).astype(np.float32)).cuda()
W0 = (rng.standard_normal((64, 1, 3, 3)) * 0.5).astype(np.float32)
def run(wmid, ch, mode):
    w = {'n_layers': np.int64(3), 'conv0.weight': W0, 'conv1.weight': wmid}
    wl = np.zeros((1, 64, 3, 3), np.float32); wl[0, ch, 1, 1] = 1.0
    w['conv2.weight'] = wl
    return ops.DncnnPlan(w, H, W, 1, winograd=mode).forward(x).cpu().numpy()[0]
rnd = (rng.standard_normal((64, 64, 3, 3)) * 0.05).astype(np.float32)
def errs(wm, chans):
    out = []
    for ch in chans:
        e = np.abs(run(wm, ch, 5) - run(wm, ch, 0)).max()
        out.append('%d:%s' % (ch, 'ok' if e < 1e-4 else '%.2f' % e))
    return ' '.join(out)
chans = [0, 15, 16, 31, 32, 33, 40, 47, 48, 55, 63]
print('dense         ', errs(rnd, chans), flush=True)
ct = np.zeros_like(rnd); ct[:, :, 1, 1] = rnd[:, :, 1, 1]
print('center tap    ', errs(ct, chans), flush=True)
for k0 in (0, 3, 4, 7):
    wm = np.zeros_like(rnd); wm[:, 8 * k0:8 * k0 + 8] = rnd[:, 8 * k0:8 * k0 + 8]
    print(f'cin chunk {k0}   ', errs(wm, chans), flush=True)
for c0 in (0, 1, 2, 7):
    wm = np.zeros_like(rnd); wm[:, c0::8] = rnd[:, c0::8]
    print(f'cin %8 == {c0}  ', errs(wm, chans), flush=True)
wm = np.zeros_like(rnd); wm[:, 1::8] = rnd[:, 1::8]
for ch in (15, 14):
    e = np.abs(run(wm, ch, 5) - run(wm, ch, 0))
    np.set_printoptions(linewidth=250, precision=2, suppress=True)
    print('ch', ch, 'error by (y % 8, x % 4):'); print(e.reshape(8, 8, 16, 4).max(axis=(0, 2)))
    print('error by (y // 8, x // 4):'); print(e.reshape(8, 8, 16, 4).max(axis=(1, 3)))
for nk in (1, 2, 3, 8):
    wm = np.zeros_like(rnd); wm[:, 1:8 * nk:8] = rnd[:, 1:8 * nk:8]
    print(f'cin = 1 mod 8, first {nk} chunks', errs(wm, [15, 31]), flush=True)
wm = np.zeros_like(rnd); wm[:, 1::8, 0, 0] = rnd[:, 1::8, 0, 0]
print('cin = 1 mod 8, tap (0,0) only', errs(wm, [15, 31]), flush=True)
wm = np.zeros_like(rnd); wm[:, 1::8, 2, 2] = rnd[:, 1::8, 2, 2]
print('cin = 1 mod 8, tap (2,2) only', errs(wm, [15, 31]), flush=True)
