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
chans = [0, 3, 14, 15, 31, 47, 63]
print('VAR', os.environ.get('PNP_W44_VAR'), 'dense', errs(rnd, chans), flush=True)
for k0 in (5, 6, 7):
    wm = np.zeros_like(rnd); wm[:, 8 * k0:8 * k0 + 8] = rnd[:, 8 * k0:8 * k0 + 8]
    print(f'   cin chunk {k0}   ', errs(wm, chans), flush=True)
