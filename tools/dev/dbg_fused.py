"""development helper: the one-kernel iteration against the streaming kernels for every operand combination; where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
os.environ['PNP_CSMRI_FUSED_MIN_BATCH'] = '1'
from pnp_svrg_amd import ops
from pnp_svrg_amd.engine import CsmriBatch
B, mb = 5, 1000
batch = CsmriBatch.synthetic(B, 256, 256, 0.2, 20.0, seed=31)
pf = batch.plan
os.environ['PNP_CSMRI_FUSED_MIN_BATCH'] = '100000'
ps = ops.CsmriPlan(256, 256, B, torch.float32)
rng = np.random.default_rng(0)
z = batch.xinit.clone()
w = (batch.xinit + torch.from_numpy(0.05 * rng.standard_normal((B, 256, 256))).float().cuda()).contiguous()
mu = torch.from_numpy(1e-4 * rng.standard_normal((B, 256, 256))).float().cuda()
sel = torch.empty((1, B, 256, 8), dtype=torch.int32, device='cuda')
pf.draw_thresholds(batch.bits, mb, seed=3, step0=7, nsteps=1, selbits=sel)
lr = 2e3
for name, kw in (('none', {}), ('c1', dict(beta=0.5, c1=w)), ('c2', dict(gamma=-0.25, c2=mu)), ('c1c2', dict(beta=1.0, c1=z, gamma=-lr, c2=mu))):
    for bb in (None, w):
        for bits in (sel[0], batch.bits):
            ga = pf.grad(z, bits=bits, b=bb, alpha=-lr / mb, **kw)
            gb = ps.grad(z, bits=bits, b=bb, alpha=-lr / mb, **kw)
            d = (ga - gb).abs()
            bad = (d > 2e-6 * max(1.0, gb.abs().max().item())).nonzero()
            print(f'grad  ops={name:5s} b={"w" if bb is not None else "-"} bits={"mb" if bits is sel[0] else "mask"}: max diff {d.max().item():.3e} (max |ref| {gb.abs().max().item():.3e}), bad {bad.shape[0]}', flush=True)
            if bad.shape[0]:
                bb_ = bad.cpu().numpy()
                print('   images', np.unique(bb_[:, 0]), 'rows', np.unique(bb_[:, 1])[:24], 'cols', np.unique(bb_[:, 2])[:24])
            r1, _, _ = pf.svrg_step(z, bb, bits, alpha=-lr / mb, denoise=False, **kw)
            d = (r1 - gb).abs()
            bad = (d > 2e-6 * max(1.0, gb.abs().max().item())).nonzero()
            print(f'step1 ops={name:5s} b={"w" if bb is not None else "-"}: max diff {d.max().item():.3e}, bad {bad.shape[0]}', flush=True)
            if bad.shape[0]:
                bb_ = bad.cpu().numpy()
                print('   images', np.unique(bb_[:, 0]), 'rows', np.unique(bb_[:, 1])[:24], 'cols', np.unique(bb_[:, 2])[:24])
