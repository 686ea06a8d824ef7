"""Round-3 GPU parity: the measured kernels pinned DIRECTLY to what the reference produced, at the full length of the
measured configuration (20 outer x 10 inner iterations, eta = 2e3, mini-batch 1000, seeds 0 / 1).

* `traces256_full.npz` / `traces256.npz` (tests/golden/make_golden_r3.py, make_golden.py) hold traces of the REAL reference;
* the drop-in loop in f32 and the batched engine with the one-kernel iteration (`k_svrg_iter`, what `bench.py --workload tv`
  times) are fed the reference's own minibatches (NumPy legacy stream, seed 1) and must stay within +-0.01 dB of every
  logged PSNR (north-star tolerance; stated here) and |z - z_ref| <= 1e-3 at the end;
* the same length with the DnCNN prox against `oracle.loops.pnp_svrg` (the oracle is pinned by the fixtures above).
"""
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN, golden

from oracle import loops as ol

pytestmark = pytest.mark.gpu
IMG256 = os.path.join(GOLDEN, 'synth256.png')
PSNR_TOL = 0.01 + 1e-9          # dB
ETA, T2, MB = 2e3, 10, 1000


def _legacy_problem_and_minibatches(n_inner, **kw):
    """What the reference does on seeds 0 / 1: the CSMRI constructor's draws, then one select_mb per inner iteration
    (algorithms/pnp_svrg.py:52) from the legacy stream -> (problem, int32 [n_inner][mb] index lists)."""
    import problems as P
    np.random.seed(0)
    p = P.CSMRI(IMG256, H=256, W=256, sample_prob=0.2, snr=20., **kw)
    np.random.seed(1)
    idx = np.stack([np.flatnonzero(p.select_mb(MB)) for _ in range(n_inner)]).astype(np.int32)
    return p, idx


def _engine_rows(trace_ref, n_outer):
    """reference psnr_per_iter (1 initial entry, then per outer iteration 1 entry + T2 inner entries) -> the T2 * n_outer
    entries that follow a prox evaluation, i.e. the rows of SvrgEngine.psnr_trace()."""
    ref = np.asarray(trace_ref)
    assert len(ref) == 1 + n_outer * (T2 + 1)
    return np.concatenate([ref[1 + o * (T2 + 1) + 1: 1 + (o + 1) * (T2 + 1)] for o in range(n_outer)])


@pytest.mark.parametrize('n_outer,fixture', [(4, 'traces256.npz'), (20, 'traces256_full.npz')])
@pytest.mark.parametrize('fold', [True, False])
def test_one_kernel_iteration_engine_vs_reference_trace(n_outer, fixture, fold):
    """SvrgEngine(fused=True) -- ONE kernel per inner iteration, the outer refresh folded into the first (fold) or as launches
    of its own -- on the problem of the golden traces with the reference's own minibatches, against the true-SVRG trace the
    reference produced (`truesvrg_psnr`, `truesvrg_z`).  The batch holds the problem twice: both copies must agree bit for bit."""
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, TVProx
    g = golden(fixture)
    p, idx = _legacy_problem_and_minibatches(n_outer * T2, upload=False)
    batch = CsmriBatch.from_problems([p, p])
    eng = SvrgEngine(batch, TVProx(), ETA, T2, MB, variant='svrg', fused=True, fold_outer=fold)
    assert eng.fused
    idx_d = torch.from_numpy(np.repeat(idx[:, None, :], 2, axis=1)).cuda()
    for s in range(n_outer * T2):
        eng.step(idx_d[s])
    tr = eng.psnr_trace()
    assert tr.shape == (n_outer * T2, 2) and np.array_equal(tr[:, 0], tr[:, 1])
    ref = _engine_rows(g['truesvrg_psnr'], n_outer)
    assert np.abs(tr[:, 0] - ref).max() <= PSNR_TOL, np.abs(tr[:, 0] - ref).max()
    z = eng.z.double().cpu().numpy().reshape(2, -1)
    assert np.array_equal(z[0], z[1])
    assert np.abs(z[0] - g['truesvrg_z']).max() <= 1e-3


@pytest.mark.parametrize('variant,key', [('reference', 'svrg'), ('svrg', 'truesvrg')])
def test_dropin_loop_full_length_f32(variant, key):
    """The drop-in pnp_svrg in the production dtype over the full 200 inner iterations against the reference's trace."""
    import algorithms as A
    import denoisers as D
    import problems as P
    g = golden('traces256_full.npz')
    np.random.seed(0)
    p = P.CSMRI(IMG256, H=256, W=256, sample_prob=0.2, snr=20., dtype=torch.float32)
    np.random.seed(1)
    r = A.pnp_svrg(p, D.TVDenoiser(), ETA, 2 + 20 * 53 - (1 if variant == 'svrg' else 0), T2, MB, verbose=False,
                   converge_check=False, clock=ol.CountingClock(), variant=variant)
    ps, ref = np.array(r['psnr_per_iter']), g[f'{key}_psnr']
    assert len(ps) == len(ref) == 221
    assert np.abs(ps - ref).max() <= PSNR_TOL, np.abs(ps - ref).max()
    assert np.abs(r['z'] - g[f'{key}_z']).max() <= 1e-3


def test_dncnn_prox_full_length_vs_oracle(g_csmri):
    """BASELINE config 3 at the bench's own length: 200 inner iterations (20 x 10) of pnp_svrg (true SVRG direction) on
    256 x 256 CSMRI with the DnCNN prox (reference sigma = 15 weights), default F(4x4,3x3) conv kernel, f32 -- the drop-in
    loop AND the batched engine (streaming kernels and one-kernel gradient step) against `oracle.loops.pnp_svrg`
    (torch-CPU fp32 net) on identical seeds: +-0.01 dB on every logged PSNR."""
    import algorithms as A
    import denoisers as D
    import problems as P
    from oracle import denoise as od, problems as op
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, DnCNNProx
    wts = dict(golden('dncnn_noise15.npz'))
    img = g_csmri['r256_img']
    n_outer = 20
    tt = 2 + n_outer * 53 - 1
    np.random.seed(0)
    po = op.CSMRI(None, H=256, W=256, sample_prob=0.2, snr=20., img=img)
    np.random.seed(1)
    ro = ol.pnp_svrg(po, od.DnCNNDenoiser(wts, 15), ETA, tt, T2, MB, converge_check=False, clock=ol.CountingClock(), variant='svrg')
    pso = np.array(ro['psnr_per_iter'])
    assert len(pso) == 1 + n_outer * (T2 + 1) and pso.max() > pso[0] + 3.0
    # drop-in loop
    np.random.seed(0)
    p = P.CSMRI(None, H=256, W=256, sample_prob=0.2, snr=20., img=img, dtype=torch.float32)
    np.random.seed(1)
    r = A.pnp_svrg(p, D.RealSN_DnCNNDenoiser('DnCNN', 15, weights=wts), ETA, tt, T2, MB, verbose=False, converge_check=False,
                   clock=ol.CountingClock(), variant='svrg')
    ps = np.array(r['psnr_per_iter'])
    assert len(ps) == len(pso) and np.abs(ps - pso).max() <= PSNR_TOL, np.abs(ps - pso).max()
    assert np.abs(r['z'] - ro['z']).max() <= 1e-3
    # batched engine on the reference's minibatches: streaming kernels (what B = 120 takes) and the one-kernel step
    np.random.seed(0)
    ph = P.CSMRI(None, H=256, W=256, sample_prob=0.2, snr=20., img=img, upload=False)
    np.random.seed(1)
    idx = np.stack([np.flatnonzero(ph.select_mb(MB)) for _ in range(n_outer * T2)]).astype(np.int32)
    idx_d = torch.from_numpy(idx[:, None, :]).cuda()
    ref = _engine_rows(pso, n_outer)
    for fused in (False, True):
        eng = SvrgEngine(CsmriBatch.from_problems([ph]), DnCNNProx(wts, 15), ETA, T2, MB, variant='svrg', fused=fused)
        for s in range(n_outer * T2):
            eng.step(idx_d[s])
        tr = eng.psnr_trace()[:, 0]
        assert np.abs(tr - ref).max() <= PSNR_TOL, (fused, np.abs(tr - ref).max())
        assert np.abs(eng.z.double().cpu().numpy().ravel() - ro['z']).max() <= 1e-3
