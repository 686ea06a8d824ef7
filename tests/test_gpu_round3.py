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


def test_timed_path_device_draws_vs_oracle():
    """What `bench.py --workload tv` times -- device-drawn minibatches (`k_draw_thr` -> bit-packed selectors) and ONE kernel per
    outer iteration (`k_svrg_outer`) at a batch that takes the one-kernel path (B = 192) -- against the oracle loop fed the
    very minibatches the device drew (decoded from the selector bits): +-0.01 dB on every logged PSNR and |z - z_oracle| <= 1e-3,
    for items that share a problem (each item draws its own minibatches) and for three different sampling ratios."""
    import problems as P
    from oracle import denoise as od, problems as op
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, TVProx
    n_outer, B = 3, 192
    ratios = (0.2, 0.3, 0.5)
    probs = []
    for k, a in enumerate(ratios):
        np.random.seed(10 + k)
        probs.append(P.CSMRI(IMG256, H=256, W=256, sample_prob=a, snr=20., upload=False))
    batch = CsmriBatch.from_problems([probs[i % 3] for i in range(B)])
    eng = SvrgEngine(batch, TVProx(), ETA, T2, MB, variant='svrg', seed=5)
    assert eng.fused and eng.outer_kernel_ok()
    sel = []
    for _ in range(n_outer):
        eng.run_outer(1)                                        # one draw launch + one k_svrg_outer launch
        sel.append(eng.mbs.selbits.cpu().numpy().copy())        # [T2][B][kx][ky >> 5] words of this outer iteration
    tr = eng.psnr_trace()
    assert tr.shape == (n_outer * T2, B)
    z = eng.z.double().cpu().numpy().reshape(B, -1)
    shifts = np.arange(32, dtype=np.uint32)
    seen = []
    for item in (0, 3, 1, 2, 191):
        k = item % 3
        mbs = []
        for o in range(n_outer):
            for j in range(T2):
                w = sel[o][j][item].view(np.uint32)                                  # [256 kx][8]
                ind = ((w[:, :, None] >> shifts) & 1).reshape(256, 256).T.astype(int)  # [ky][kx] = the H x W indicator select_mb returns
                assert ind.sum() == MB and (ind <= probs[k].mask).all()
                mbs.append(ind)
        seen.append(np.stack(mbs[:2]))
        np.random.seed(10 + k)
        po = op.CSMRI(IMG256, H=256, W=256, sample_prob=ratios[k], snr=20.)
        assert np.array_equal(po.mask, probs[k].mask)
        it = iter(mbs)
        po.select_mb = lambda size: next(it)
        ro = ol.pnp_svrg(po, od.TVDenoiser(), ETA, 2 + n_outer * 53 - 1, T2, MB, converge_check=False, clock=ol.CountingClock(), variant='svrg')
        ref = _engine_rows(ro['psnr_per_iter'], n_outer)
        assert np.abs(tr[:, item] - ref).max() <= PSNR_TOL, (item, np.abs(tr[:, item] - ref).max())
        assert np.abs(z[item] - ro['z']).max() <= 1e-3, item
    assert not np.array_equal(seen[0], seen[1])                 # items 0 and 3 share a problem, not their minibatches


def test_timed_path_dncnn_device_draws_vs_oracle(g_csmri):
    """The same for what the default `bench.py` line times (config 3): device-drawn minibatches, the streaming CSMRI kernels a
    batch below 192 takes, the DnCNN prox with the default F(4x4,3x3) conv kernel -- two outer iterations of a small batch
    against the oracle loop (torch-CPU fp32 net) fed the decoded device draws: +-0.01 dB, |z - z_oracle| <= 1e-3."""
    import problems as P
    from oracle import denoise as od, problems as op
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, DnCNNProx
    wts = dict(golden('dncnn_noise15.npz'))
    img = g_csmri['r256_img']
    n_outer, B = 2, 4
    np.random.seed(21)
    ph = P.CSMRI(None, H=256, W=256, sample_prob=0.2, snr=20., img=img, upload=False)
    batch = CsmriBatch.from_problems([ph] * B)
    eng = SvrgEngine(batch, DnCNNProx(wts, 15), ETA, T2, MB, variant='svrg', seed=11)
    assert not eng.fused
    sel = []
    for _ in range(n_outer):
        for _ in range(T2):
            eng.step()                                          # device draws: T2 selections per outer iteration
        sel.append(eng.mbs.selbits.cpu().numpy().copy())
    tr = eng.psnr_trace()
    z = eng.z.double().cpu().numpy().reshape(B, -1)
    shifts = np.arange(32, dtype=np.uint32)
    for item in (0, 3):
        mbs = [((sel[o][j][item].view(np.uint32)[:, :, None] >> shifts) & 1).reshape(256, 256).T.astype(int)
               for o in range(n_outer) for j in range(T2)]
        assert all(m.sum() == MB for m in mbs)
        np.random.seed(21)
        po = op.CSMRI(None, H=256, W=256, sample_prob=0.2, snr=20., img=img)
        it = iter(mbs)
        po.select_mb = lambda size: next(it)
        ro = ol.pnp_svrg(po, od.DnCNNDenoiser(wts, 15), ETA, 2 + n_outer * 53 - 1, T2, MB, converge_check=False, clock=ol.CountingClock(),
                         variant='svrg')
        ref = _engine_rows(ro['psnr_per_iter'], n_outer)
        assert np.abs(tr[:, item] - ref).max() <= PSNR_TOL, (item, np.abs(tr[:, item] - ref).max())
        assert np.abs(z[item] - ro['z']).max() <= 1e-3, item
    assert not np.array_equal(tr[:, 0], tr[:, 3]) or not np.array_equal(z[0], z[3])      # different draws, different paths


# ------------------------------------------------------------------------------------------------ general sweep runner
def _smooth_images(n, count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        x = rng.random((n, n))
        p = np.pad(x, 2, mode='wrap')
        out.append(sum(p[i:i + n, j:j + n] for i in range(5) for j in range(5)) / 25.0)
    return out


def test_full_size_batches_items_do_not_depend_on_the_batch():
    """The benchmarked batch sizes themselves (config 2: B = 1024 through `k_svrg_outer`; config 3: B = 120 through the streaming
    kernels + DnCNN; config 4: B = 64 Deblur + NLM + pnp_saga): a reconstruction never sees its neighbours, and device draws are keyed by (seed, step, problem), so the first
    items of the full-size batch must equal, BIT FOR BIT, the same items run in a small batch -- iterate and PSNR log."""
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, TVProx, DnCNNProx
    wts = dict(golden('dncnn_noise15.npz'))
    for (B_full, B_small, mk_prox, n_outer) in ((1024, 192, lambda: TVProx(), 2), (120, 8, lambda: DnCNNProx(wts, 15), 1)):
        big = CsmriBatch.synthetic(B_full, 256, 256, 0.2, 20.0, seed=100)
        small = CsmriBatch.synthetic(B_small, 256, 256, 0.2, 20.0, seed=100)      # (one Generator stream, problem after problem: the same first problems)
        assert torch.equal(small.xrec, big.xrec[:B_small]) and torch.equal(small.bits, big.bits[:B_small])
        out = []
        for batch in (big, small):
            eng = SvrgEngine(batch, mk_prox(), ETA, T2, MB, variant='svrg', seed=1)
            if eng.outer_kernel_ok():
                eng.run_outer(n_outer)
            else:
                for _ in range(n_outer * T2):
                    eng.step()
            out.append((eng.z[:B_small].clone(), eng.psnr_trace()[:, :B_small].copy()))
            del eng
        assert torch.equal(out[0][0], out[1][0]), B_full
        assert np.array_equal(out[0][1], out[1][1]), B_full
        del big, small, out
        torch.cuda.empty_cache()
    # config 4: B = 64 Deblur problems, pnp_saga + NLM on device draws
    from pnp_svrg_amd.engine import DeblurBatch, NLMProx, make_engine
    out = []
    big = DeblurBatch.synthetic(64, 256, 256, 'Minimal', 20.0, seed=100)
    Bk = np.zeros((256, 256))
    Bk[0, 0] = Bk[128, 128] = Bk[128, 85] = Bk[128, 64] = 0.25                 # the "Minimal" kernel of DeblurBatch.synthetic (DeblurSR.py:80-89)
    small = DeblurBatch(big.xrec[:4].double().cpu().numpy(), Bk.ravel() / 65536, big.Y[:4].double().cpu().numpy(),
                        big.xinit[:4].double().cpu().numpy().reshape(4, -1))
    for batch in (big, small):
        eng = make_engine(batch, NLMProx(), 5e6, 1, 3000, algorithm='saga', hist_size=50, seed=1)
        for _ in range(3):
            eng.step()
        out.append((batch.xrec[:4].clone(), eng.z[:4].clone(), eng.psnr_trace()[:, :4].copy()))
        del eng
    del big, small
    torch.cuda.empty_cache()
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize('algo', ['sgd', 'sarah'])
def test_per_step_engines_device_draws_vs_oracle(algo):
    """SgdEngine / SarahEngine on DEVICE-drawn minibatches (a window of steps per draw launch) against the oracle loops fed the
    same selections -- the draw is a function of (seed, step, problem), re-drawn here step by step as bit-packed selectors and
    decoded -- in float64: identical PSNR logs, |z - z_oracle| <= 1e-9.  Three sampling ratios in one batch."""
    import problems as P
    from oracle import denoise as od, problems as op
    from pnp_svrg_amd.engine import CsmriBatch, TVProx, make_engine
    img64 = os.path.join(GOLDEN, 'synth64.png')
    n, mb, T2_, eta, seed, decay = 64, 150, 4, 5e2, 3, 0.9
    steps = 2 * T2_ + 1 if algo == 'sarah' else 20             # (sgd: more steps than one draw window holds)
    ratios = (0.3, 0.5, 0.8)
    probs = []
    for k, a in enumerate(ratios):
        np.random.seed(30 + k)
        probs.append(P.CSMRI(img64, H=n, W=n, sample_prob=a, snr=20., dtype=torch.float64, upload=False))
    batch = CsmriBatch.from_problems(probs, dtype=torch.float64)
    eng = make_engine(batch, TVProx(), eta, T2_, mb, algorithm=algo, seed=seed, lr_decay=decay)
    for _ in range(steps):
        eng.step()
    tr = eng.psnr_trace()
    z = eng.z.cpu().numpy().reshape(batch.B, -1)
    shifts = np.arange(32, dtype=np.uint32)
    sb = torch.zeros((1, batch.B, n, n // 32), dtype=torch.int32, device='cuda')
    sel = []
    for st in range(steps):
        batch.plan.draw_thresholds(batch.bits, mb, seed, st, 1, selbits=sb)
        sel.append(sb.cpu().numpy().copy()[0])
    for b, a in enumerate(ratios):
        mbs = [((sel[st][b].view(np.uint32)[:, :, None] >> shifts) & 1).reshape(n, n).T.astype(int) for st in range(steps)]
        assert all(m.sum() == mb and (m <= probs[b].mask).all() for m in mbs)
        np.random.seed(30 + b)
        po = op.CSMRI(img64, H=n, W=n, sample_prob=a, snr=20.)
        it = iter(mbs)
        po.select_mb = lambda size: next(it)
        kw = dict(converge_check=False, clock=ol.CountingClock(), lr_decay=decay)
        if algo == 'sgd':
            ro = ol.pnp_sgd(po, od.TVDenoiser(), eta, 5 * steps - 2, mb, **kw)
            ref = np.array(ro['psnr_per_iter'])[1:]
        else:
            o, j = (steps - 1) // T2_, (steps - 1) % T2_
            ro = ol.pnp_sarah(po, od.TVDenoiser(), eta, 1 + o * (5 + 5 * T2_) + 5 + 5 * j + 1, T2_, mb, **kw)
            ref = np.array(ro['psnr_per_iter'])                  # outer prox entries included, like the engine's log
        assert len(ref) == tr.shape[0], (len(ref), tr.shape)
        assert np.abs(tr[:, b] - ref).max() <= 1e-9, (algo, b)
        assert np.abs(z[b] - ro['z']).max() <= 1e-9


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_timed_path_saga_nlm_device_draws_vs_oracle(dtype):
    """Config 4's timed path (`bench.py --workload saga-nlm`): SagaEngine over a DeblurBatch with DEVICE-drawn minibatches
    (threshold descriptors read by the gradient kernel) and the NLM prox, against `oracle.loops.pnp_saga` fed the same selections
    (the draw is a function of (seed, step, problem): materialised here with `indicator_from_thresholds`) and the same replaced
    table rows: f64 identical, f32 +-0.01 dB."""
    import problems as P
    from pnp_svrg_amd import ops
    from pnp_svrg_amd.engine import DeblurBatch, NLMProx, make_engine
    from oracle import denoise as od, problems as op
    n, mb, hist, n_it, eta, seed = 64, 300, 4, 6, 1e7, 5
    imgs = _smooth_images(n, 2, 3)
    spec = [(0, 0, 20.0), (5, 1, 20.0), (2, 0, 10.0)]                     # (constructor seed, image, snr)
    probs = []
    for sd, im, snr in spec:
        np.random.seed(sd)
        probs.append(P.Deblur(None, H=n, W=n, kernel='Minimal', scale_percent=100, snr=snr, img=imgs[im], dtype=dtype, upload=False))
    batch = DeblurBatch.from_problems(probs, dtype=dtype)
    B = batch.B
    eng = make_engine(batch, NLMProx(), eta, 1, mb, algorithm='saga', hist_size=hist, seed=seed)
    rs = np.random.RandomState(1)                                         # the legacy stream the oracle loop draws its rows from
    rows = [int(rs.choice(hist, 1).item()) for _ in range(n_it)]
    for k in range(n_it):
        eng.step(r=rows[k])                                               # device draws; the replaced row imposed
    tr = eng.psnr_trace()
    z = eng.z.double().cpu().numpy().reshape(B, -1)
    # the selections the device used: the table-filling draw (step id 0xFFFFFFFF), then steps 0 .. n_it - 1
    steps = [0xFFFFFFFF] + list(range(n_it))
    ind = [ops.indicator_from_thresholds(batch.M, ops.draw_thresholds(batch.M, B, mb, seed, st, 1)[0]).cpu().numpy() for st in steps]
    for b, (sd, im, snr) in enumerate(spec):
        np.random.seed(sd)
        po = op.Deblur(None, H=n, W=n, kernel='Minimal', scale_percent=100, snr=snr, img=imgs[im])
        mbs = [np.asarray(i[b]).reshape(-1).astype(int) for i in ind]
        assert all(m.sum() == mb for m in mbs)
        it = iter(mbs)
        po.select_mb = lambda size: next(it)
        d = od.NLMDenoiser()
        d.sigma = 1.0
        np.random.seed(1)
        ro = ol.pnp_saga(po, d, eta, 5 * n_it - 1, mb, hist_size=hist, converge_check=False, clock=ol.CountingClock())
        ref = np.array(ro['psnr_per_iter'])
        assert len(ref) == n_it + 1
        if dtype == torch.float64:
            assert np.array_equal(tr[:, b], ref[1:]) and np.abs(z[b] - ro['z']).max() <= 1e-9
        else:
            assert np.abs(tr[:, b] - ref[1:]).max() <= PSNR_TOL and np.abs(z[b] - ro['z']).max() <= 1e-3


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_sweep_runner_deblur_nlm_saga_vs_oracle(dtype):
    """One cell of the reference's sweep beyond CSMRI (script_diff_sampratio_set12.py:23-25: DeblurSR x NLM x pnp_saga): three
    legacy-seeded items in ONE batch through `sweep.make_runner`, each against `oracle.loops.pnp_saga` on the same seeds
    (constructor draws, then per iteration select_mb + np.random.choice(hist_size, 1): every item its own stream)."""
    from pnp_svrg_amd import sweep
    from oracle import denoise as od, problems as op
    n, mb, hist, n_it, eta = 64, 300, 4, 5, 1e7           # (the reference's blur has gain 1 / sqrt(N): the step is stable below ~ 2 N^2 = 3e7)
    imgs = _smooth_images(n, 2, 3)
    items = [{'id': 0, 'image': 0, 'alpha': 1.0, 'snr': 20.0, 'seed': 0}, {'id': 1, 'image': 1, 'alpha': 1.0, 'snr': 20.0, 'seed': 5},
             {'id': 2, 'image': 0, 'alpha': 1.0, 'snr': 10.0, 'seed': 2}]
    run = sweep.make_runner(imgs, 'deblur', 'saga', 'nlm', eta=eta, n_inner=n_it, mini_batch_size=mb, hist_size=hist, H=n, W=n,
                            dtype=dtype, seeding='legacy', keep_trace=True)
    assert run.names == ('DeblurSR', 'NLM', 'pnp_saga')
    res = sweep.run_sweep(items, run)
    for it, r in zip(items, res):
        np.random.seed(it['seed'])
        p = op.Deblur(None, H=n, W=n, kernel='Minimal', scale_percent=100, snr=it['snr'], img=imgs[it['image']])
        d = od.NLMDenoiser()
        d.sigma = 1.0
        np.random.seed(1)
        ro = ol.pnp_saga(p, d, eta, 5 * n_it - 1, mb, hist_size=hist, converge_check=False, clock=ol.CountingClock())
        ref = np.array(ro['psnr_per_iter'])
        assert len(ref) == n_it + 1 and r['psnr_init'] == ref[0]
        if dtype == torch.float64:
            assert np.array_equal(r['psnr_trace'], ref[1:]) and np.isfinite(ref).all()
            assert np.abs(r['z'].ravel() - ro['z']).max() <= 1e-9
        else:
            assert np.abs(r['psnr_trace'] - ref[1:]).max() <= PSNR_TOL
            assert np.abs(r['z'].ravel() - ro['z']).max() <= 1e-3
    assert len({tuple(r['psnr_trace']) for r in res}) == 3                  # three different problems


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_sweep_runner_pr_tv_sarah_vs_oracle(dtype):
    """PR (32 x 32, num_meas = alpha * 1024 as script_diff_sampratio_set12.py:47-48) x TV x pnp_sarah through `sweep.make_runner`:
    items of two oversampling ratios (two batches, grouped by alpha), legacy seeding, each against `oracle.loops.pnp_sarah`."""
    from pnp_svrg_amd import sweep
    from oracle import denoise as od, problems as op
    n, mb, T2, n_it, eta = 32, 200, 3, 7, 0.05
    imgs = _smooth_images(n, 2, 7)
    items = [{'id': 0, 'image': 0, 'alpha': 2.0, 'snr': 20.0, 'seed': 0}, {'id': 1, 'image': 1, 'alpha': 3.0, 'snr': 20.0, 'seed': 1},
             {'id': 2, 'image': 1, 'alpha': 2.0, 'snr': 20.0, 'seed': 4}]
    run = sweep.make_runner(imgs, 'pr', 'sarah', 'tv', eta=eta, n_inner=n_it, mini_batch_size=mb, T2=T2, H=n, W=n, dtype=dtype,
                            seeding='legacy', keep_trace=True, lr_decay=0.9)
    res = sweep.run_sweep(items, run)
    assert [r['id'] for r in res] == [0, 1, 2]
    o, j = (n_it - 1) // T2, (n_it - 1) % T2
    for it, r in zip(items, res):
        np.random.seed(it['seed'])
        p = op.PhaseRetrieval(None, H=n, W=n, num_meas=int(it['alpha'] * n * n), snr=it['snr'], img=imgs[it['image']])
        np.random.seed(1)
        ro = ol.pnp_sarah(p, od.TVDenoiser(), eta, 1 + o * (5 + 5 * T2) + 5 + 5 * j + 1, T2, mb, converge_check=False,
                          clock=ol.CountingClock(), lr_decay=0.9)
        ref = np.array(ro['psnr_per_iter'])                      # outer prox entries included, like the engine's log
        assert len(ref) == len(r['psnr_trace'])
        if dtype == torch.float64:
            assert np.abs(r['psnr_trace'] - ref).max() <= 1e-9
            assert np.abs(r['z'].ravel() - ro['z']).max() <= 1e-8
        else:
            assert np.abs(r['psnr_trace'] - ref).max() <= PSNR_TOL
            assert np.abs(r['z'].ravel() - ro['z']).max() <= 2e-3


def test_sweep_runner_cells_and_hist_size_grid(tmp_path):
    """Every (problem, algorithm) cell steps through `make_runner` with generator seeding and device draws (NLM with pnp_svrg
    included: it cannot be captured in a hipGraph and must fall back to eager stepping -- ADVICE r2), and `hist_size` is a
    searchable key of `grid_search` for pnp_saga (script_diff_snr_set12.py search space)."""
    from pnp_svrg_amd import sweep
    n = 64
    imgs = _smooth_images(n, 2, 11)
    items = sweep.make_items(2, [1.0], [20.0])
    for problem, eta, mb in (('csmri', 5e2, 100), ('deblur', 1e7, 300)):
        for algo in sweep.ALGORITHMS:
            its = items if problem == 'deblur' else sweep.make_items(2, [0.3, 0.5], [20.0])
            run = sweep.make_runner(imgs, problem, algo, 'tv', eta=eta, n_inner=8, mini_batch_size=mb, T2=4, hist_size=3, H=n, W=n)
            res = sweep.run_sweep(its, run)
            assert len(res) == len(its) and all(np.isfinite(r['psnr_final']) for r in res), (problem, algo)
    # NLM prox under pnp_svrg with device draws: 8 = 2 x T2 inner iterations would take the graph path if it could
    run = sweep.make_runner(imgs, 'csmri', 'svrg', 'nlm', eta=5e2, n_inner=8, mini_batch_size=100, T2=4, H=n, W=n)
    res = sweep.run_sweep(sweep.make_items(2, [0.4], [20.0]), run)
    assert all(np.isfinite(r['psnr_final']) for r in res)
    its = [{'id': 0, 'image': 0, 'alpha': 2.0, 'snr': 20.0, 'seed': 0}, {'id': 1, 'image': 1, 'alpha': 2.0, 'snr': 20.0, 'seed': 0}]
    pr_imgs = _smooth_images(32, 2, 5)
    res = sweep.run_sweep(its, sweep.make_runner(pr_imgs, 'pr', 'svrg', 'tv', eta=0.05, n_inner=6, mini_batch_size=200, T2=3, H=32, W=32))
    assert all(np.isfinite(r['psnr_final']) for r in res)

    def mk(eta, hist_size):
        return sweep.make_runner(imgs, 'deblur', 'saga', 'tv', eta=eta, n_inner=6, mini_batch_size=300, hist_size=hist_size, H=n, W=n)
    rows = sweep.grid_search(items, mk, {'eta': [5e6, 1e7], 'hist_size': [2, 5]})
    assert [r['id'] for r in rows] == [0, 1] and all(set(r['params']) == {'eta', 'hist_size'} for r in rows)
    sweep.write_tuning_csv(str(tmp_path / 't.csv'), rows, problem='DeblurSR', denoiser='TV', algorithm='pnp_saga')
    txt = (tmp_path / 't.csv').read_text().splitlines()
    assert txt[0] == 'Results:' and txt[1].startswith('DeblurSR,TV,pnp_saga,1.0,20.0,') and ',PARAMETERS:,eta,' in txt[1] and ',hist_size,' in txt[1]


# ------------------------------------------------------------------------------------------------ config 5 bench path
def test_bench_sweep_workload_one_gpu():
    """`bench.py --workload sweep` (BASELINE config 5: 120 work items through `sweep` with the gather inside the clock) on one
    GPU: contract fields, strong scaling, every item gathered, PSNR improved."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--workload', 'sweep', '--steps', '10', '--warmup', '1',
                          '--no-cpu-baseline'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['scaling'] == 'strong' and line['n_gpus'] == 1 and line['unit'] == 'item-inner-iters/s'
    assert line['config']['items_total'] == 120 and line['psnr_db']['items_gathered'] == 120
    assert line['warmup'] == 10 and line['steps'] == 10               # whole outer iterations (hipGraph replays)
    assert abs(line['value'] - 120 * 10 / (line['ms_per_step'] * 10 / 1e3)) < 1e-2 * line['value']
    assert 0 < line['roofline']['frac'] < 1 and line['psnr_db']['after_timed_steps_mean'] > 15


def test_bench_sweep_two_ranks_rehearsal():
    """The N > 1 path of the sweep workload (items dealt round-robin, one batch per rank, tensor gather inside the clock, MAX
    over ranks) with 2 ranks sharing GPU 0 over gloo (RCCL needs one GPU per rank; the driver runs the real N = 2/4/8)."""
    import json, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PNP_BENCH_ONE_DEVICE='1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'),
                          '--gpus', '2', '--workload', 'sweep', '--steps', '10', '--warmup', '10', '--backend', 'gloo',
                          '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['config']['items_per_gpu'] == 60
    assert line['psnr_db']['items_gathered'] == 120 and line['value'] > 0
