"""Round-2 parity tests on the GPU: the fixtures of tests/golden/make_golden_r2.py / make_golden_dncnn_r2.py (reference
outputs), BASELINE configs 4 and 5 at full size, and the batched engines over Deblur / phase-retrieval batches."""
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN, golden

from oracle import problems as op, loops as ol, denoise as od

pytestmark = pytest.mark.gpu
IMG256 = os.path.join(GOLDEN, 'synth256.png')
IMG64 = os.path.join(GOLDEN, 'synth64.png')


def _png(tmp_path, name, pixels):
    from PIL import Image
    path = str(tmp_path / name)
    Image.fromarray(pixels).save(path)
    return path


def feed_minibatches(p, lists, shape):
    it = iter(lists)

    def select_mb(size, _it=it):
        m = np.zeros(int(np.prod(shape)), int)
        m[next(_it)] = 1
        return m.reshape(shape)
    p.select_mb = select_mb


def test_resize_and_kernel_png(g_r2, tmp_path):
    """a6 / a13 through the product classes: Problem(img_path) resizes the 512^2 photograph like the reference (PIL
    bicubic + min-max), Deblur(kernel_path=...) builds the reference's blur vector from kernel.png, and the device
    gradient on that kernel matches the reference's."""
    import problems
    g = g_r2
    p = problems.Problem(_png(tmp_path, '08.png', g['resize_pixels']), 256, 256)
    assert np.array_equal(p.Xrec, g['resize_Xrec'])
    kp = _png(tmp_path, 'kernel.png', g['kernelpng_pixels'])
    np.random.seed(0)
    p64 = problems.Deblur(IMG64, H=64, W=64, kernel_path=kp, scale_percent=100, snr=20., dtype=torch.float64)
    assert np.array_equal(p64.B, g['kernelpng_B64'])
    for dtype, rel in ((torch.float64, 1e-10), (torch.float32, 2e-4)):
        np.random.seed(0)
        p = problems.Deblur(IMG256, H=256, W=256, kernel_path=kp, scale_percent=100, snr=20., dtype=dtype)
        assert np.array_equal(p.B, g['kernelpng_B256'])
        assert p.sigma == pytest.approx(float(g['kernelpng_sigma256']), rel=1e-12 if dtype == torch.float64 else 3e-5)
        np.testing.assert_array_equal(p.Xinit, g['kernelpng_Xinit256'])
        gf = g['kernelpng_grad_full256']
        assert np.abs(p.grad_full(g['kernelpng_Xinit256']) - gf).max() <= rel * np.abs(gf).max()


def test_psnr_rounding_edge_on_device(g_psnr):
    """a7: Problem.PSNR (device error sum + host rounding) on the fixture that sits on a rounding boundary."""
    import problems
    g = g_psnr
    p = problems.Problem(None, 64, 64, img=g['Xrec'], dtype=torch.float64)
    assert np.array_equal(p.Xrec, g['Xrec'])
    got = [p.PSNR(w) for w in g['ws']]
    assert got == list(g['psnr'])
    got_dev = [p.PSNR(p.to_device(w)) for w in g['ws']]
    assert got_dev == list(g['psnr'])


def test_nlm_256(g_r2, g_denoise):
    """a19 at BASELINE's size: bit-exact in float64, float32 within 2e-6 of it."""
    import denoisers
    z0 = g_denoise['r256_z0']
    s = float(g_r2['nlm256_sigma_est'])
    d = denoisers.NLMDenoiser(dtype=torch.float64)
    d.sigma = 1.0
    assert np.array_equal(d.denoise(noisy=z0, sigma_est=s), g_r2['nlm256_out'])
    d32 = denoisers.NLMDenoiser(dtype=torch.float32)
    d32.sigma = 1.0
    assert np.abs(d32.denoise(noisy=z0, sigma_est=s) - g_r2['nlm256_out']).max() <= 5e-6
    # the noise estimate the loop feeds it
    from pnp_svrg_amd import ops
    zt = torch.from_numpy(z0).cuda().reshape(1, 256, 256)
    assert abs(float(ops.sigma_est(zt)[0]) - s) <= 1e-15


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_config4_full_size(g_r2, dtype):
    """BASELINE config 4 at 256 x 256: Deblur ("Minimal") + NLM prox + pnp_saga, drop-in loop against the REFERENCE's
    own trace (identical RNG stream); then the batched SagaEngine over a DeblurBatch against that drop-in loop."""
    import algorithms, problems, denoisers
    from pnp_svrg_amd.engine import DeblurBatch, NLMProx, make_engine
    g = g_r2
    np.random.seed(0)
    p = problems.Deblur(IMG256, H=256, W=256, kernel='Minimal', scale_percent=100, snr=20., dtype=dtype)
    f64 = dtype == torch.float64
    assert p.sigma == pytest.approx(float(g['c4_sigma']), rel=1e-12 if f64 else 3e-5)
    np.random.seed(3)
    mb = p.select_mb(3000)
    assert np.array_equal(mb, g['c4_mb'])
    rel = 1e-10 if f64 else 2e-4
    for got, key in ((p.grad_full(g['c4_Xinit']), 'c4_grad_full'), (p.grad_stoch(g['c4_Xinit'], mb), 'c4_grad_stoch')):
        assert np.abs(got - g[key]).max() <= rel * np.abs(g[key]).max()
    np.random.seed(1)
    d = denoisers.NLMDenoiser(dtype=dtype)
    d.sigma = 1.0
    r = algorithms.pnp_saga(p, d, 1e9, 5 * 6 - 1, 3000, hist_size=4, verbose=False, converge_check=False, clock=ol.CountingClock())
    ps = np.array(r['psnr_per_iter'])
    if f64:
        assert list(ps) == list(g['c4_saga_nlm_psnr'])
        np.testing.assert_allclose(r['z'], g['c4_saga_nlm_z'], rtol=0, atol=1e-8)
    else:
        assert len(ps) == len(g['c4_saga_nlm_psnr']) and np.abs(ps - g['c4_saga_nlm_psnr']).max() <= 0.01 + 1e-9
    # engine: the same problem twice in a batch, fed the minibatch / replaced-row stream the loop above consumed
    np.random.seed(1)
    lists, rs = [np.flatnonzero(problems.Problem.select_mb(p, 3000))], []
    for _ in range(6):
        lists.append(np.flatnonzero(problems.Problem.select_mb(p, 3000)))
        rs.append(np.random.choice(4, 1).item())
    batch = DeblurBatch.from_problems([p, p], dtype=dtype)
    idx = torch.from_numpy(np.stack([np.stack([l, l]) for l in lists]).astype(np.int32)).cuda()
    eng = make_engine(batch, NLMProx(), 1e9, 1, 3000, algorithm='saga', hist_size=4, idx0=idx[0])
    for s in range(6):
        eng.step(idx[s + 1], r=rs[s])
    tr = eng.psnr_trace()
    for b in range(2):
        assert np.abs(tr[:, b] - ps[1:]).max() <= (1e-9 if f64 else 0.01 + 1e-9)
        assert np.abs(eng.z[b].double().cpu().numpy().ravel() - r['z']).max() <= (1e-9 if f64 else 1e-3)   # (eta = 1e9 amplifies f32 rounding)


def test_saga_engine_device_draws_deblur():
    """DeblurBatch with device-drawn minibatches: the gradient with the threshold descriptor == the gradient with the
    materialised indicator; a SAGA + NLM run reconstructs (PSNR rises) and is deterministic in its seed."""
    from pnp_svrg_amd import ops
    from pnp_svrg_amd.engine import DeblurBatch, NLMProx, make_engine
    batch = DeblurBatch.synthetic(3, 64, 64, 'Minimal', 20.0, seed=2)
    mbd = ops.draw_thresholds(batch.M, batch.B, 500, seed=9, step0=4)[0]
    sel = ops.indicator_from_thresholds(batch.M, mbd)
    z = batch.xinit.clone()
    g1 = batch.plan.grad(z, batch.Y, sel=sel, scale=1.0 / 500)
    g2 = batch.plan.grad(z, batch.Y, mbd=mbd, scale=1.0 / 500)
    assert torch.equal(g1, g2) and float(g1.abs().max()) > 0
    runs = []
    for _ in range(2):
        eng = make_engine(batch, NLMProx(), 1e5, 1, 500, algorithm='saga', hist_size=6, seed=5)
        for _ in range(8):
            eng.step()
        runs.append((eng.z.clone(), eng.psnr_trace()))
    assert torch.equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert (runs[0][1][-1] > batch.psnr_init()).all()


@pytest.mark.parametrize('algo', ['svrg', 'sgd'])
def test_pr_batch_engine_matches_dropin(algo):
    """PrBatch (B phase-retrieval problems, each with its own A; batched GEMV kernels) through SvrgEngine / SgdEngine ==
    the drop-in loops on each problem (float64, same minibatches)."""
    import algorithms, problems, denoisers
    from pnp_svrg_amd.engine import PrBatch, TVProx, make_engine
    probs = []
    for seed in (0, 4):
        np.random.seed(seed)
        probs.append(problems.PhaseRetrieval(IMG64, H=32, W=32, num_meas=3 * 1024, snr=20., dtype=torch.float64))
    batch = PrBatch.from_problems(probs, dtype=torch.float64)
    B, mb, T2, steps, eta = 2, 500, 4, 8, 0.2
    idx = batch.draw_minibatches(steps, mb, seed=6)                  # ascending row ids, like np.flatnonzero(indicator)
    eng = make_engine(batch, TVProx(), eta, T2, mb, algorithm=algo, variant='svrg')
    for s in range(steps):
        eng.step(idx[s])
    tr = eng.psnr_trace()
    idx_h = idx.cpu().numpy()
    for b in range(B):
        p = probs[b]
        feed_minibatches(p, idx_h[:, b], (p.M,))
        kw = dict(verbose=False, converge_check=False, clock=ol.CountingClock())
        if algo == 'svrg':
            r = algorithms.pnp_svrg(p, denoisers.TVDenoiser(), eta, 2 + 3 * 2 + 5 * steps - 1, T2, mb, variant='svrg', **kw)
            got = [v for i, v in enumerate(np.array(r['psnr_per_iter'])[1:]) if i % (T2 + 1) != 0]
        else:
            r = algorithms.pnp_sgd(p, denoisers.TVDenoiser(), eta, 5 * steps - 2, mb, **kw)
            got = np.array(r['psnr_per_iter'])[1:]
        assert np.abs(np.array(got) - tr[:, b]).max() <= 1e-9
        np.testing.assert_allclose(r['z'], eng.z[b].cpu().numpy().ravel(), rtol=0, atol=1e-9)
    # device draws: rows re-derived from the thresholds, run is deterministic and finite
    e2 = make_engine(batch, TVProx(), eta, T2, mb, algorithm=algo, variant='svrg', seed=3)
    for _ in range(steps):
        e2.step()
    assert np.isfinite(e2.psnr_trace()).all()


@pytest.mark.parametrize('sigma', [5, 40])
def test_dncnn_other_noise_levels(sigma, g_denoise):
    """a20 at sigma = 5 and 40 (the other weights the reference ships): the MFMA net inside the wrapper against the
    outputs of the reference class, 64 x 64 and 256 x 256."""
    import denoisers
    g = golden(f'dncnn_noise{sigma}.npz')
    w = {k: g[k] for k in g.files if k.startswith(('conv', 'bn')) or k == 'n_layers'}
    d = denoisers.RealSN_DnCNNDenoiser('DnCNN', sigma, weights=w)
    assert np.abs(d.denoise(noisy=g_denoise['s64_z0']) - g['den64']).max() <= 3e-5
    assert np.abs(d.denoise(noisy=g_denoise['r256_z0']) - g['den256']).max() <= 3e-5


def test_config5_sweep_dncnn_vs_oracle(g_csmri):
    """BASELINE config 5 on one GPU: `sweep.run_sweep` over images x sampling ratios with the DnCNN prox at 256 x 256,
    ONE mixed-alpha batch (per-problem M0), problems and minibatches drawn from the legacy np.random stream exactly as
    the reference's loop would (script_diff_sampratio_set12.py:113-146 per item) -- every item's PSNR trace within
    +-0.01 dB of the oracle's pnp_svrg on the same seeds."""
    from PIL import Image
    from pnp_svrg_amd import sweep
    from pnp_svrg_amd.engine import DnCNNProx
    wts = dict(golden('dncnn_noise15.npz'))
    images = [g_csmri['r256_img'], np.array(Image.open(IMG256))]
    alphas, T2, mb, n_inner, eta = [0.2, 0.3, 0.5], 3, 1000, 5, 2e3
    items = sweep.make_items(len(images), alphas, [20.0])
    runner = sweep.csmri_svrg_runner(images, lambda: DnCNNProx(wts, 15), eta=eta, T2=T2, mini_batch_size=mb, n_inner=n_inner,
                                     seeding='legacy', keep_trace=True)
    res = sweep.run_sweep(items, runner)
    assert [r['id'] for r in res] == list(range(6))
    assert len({r['M0'] for r in res}) >= 3                       # a genuinely mixed batch
    n_outer = -(-n_inner // T2)
    for r in res:
        it = r['item']
        np.random.seed(it['seed'])
        po = op.CSMRI(None, H=256, W=256, sample_prob=it['alpha'], snr=it['snr'], img=images[it['image']])
        assert po.M0 == r['M0']
        np.random.seed(1)
        ro = ol.pnp_svrg(po, od.DnCNNDenoiser(wts, 15), eta, 2 + 3 * n_outer + 5 * n_inner - 1, T2, mb, converge_check=False,
                         clock=ol.CountingClock(), variant='svrg')
        pso = np.array(ro['psnr_per_iter'])
        inner = np.array([v for i, v in enumerate(pso[1:]) if i % (T2 + 1) != 0])
        assert len(inner) == n_inner
        assert np.abs(inner - r['psnr_trace']).max() <= 0.01 + 1e-9, (it, inner, r['psnr_trace'])
        assert abs(r['psnr_init'] - pso[0]) <= 0.01 + 1e-9
        assert np.abs(r['z'].ravel() - ro['z']).max() < 5e-4
        assert np.isfinite(r['psnr_final'])


@pytest.mark.gpu
@pytest.mark.parametrize('algo', ['sgd', 'saga', 'sarah'])
def test_draw_ahead_window_equals_single_step_draws(algo):
    """The per-step engines draw AHEAD steps of minibatches per launch (the draw kernel is latency-bound): the same
    iterates and logs, bit for bit, as with one draw launch per step -- over a window boundary, over both batch types,
    and with a host-fed step in the middle of a window."""
    from pnp_svrg_amd import engine as E
    runs = []
    for ahead in (16, 1, 5):
        for mk in (lambda: E.CsmriBatch.synthetic(3, 64, 64, 0.25, 20.0, seed=8), lambda: E.DeblurBatch.synthetic(2, 64, 64, 'Minimal', 20.0, seed=2)):
            batch = mk()
            old = E._StochEngine.AHEAD
            E._StochEngine.AHEAD = ahead
            try:
                eng = E.make_engine(batch, E.TVProx(), 5e2 if isinstance(batch, E.CsmriBatch) else 1e3, 4, 150, algorithm=algo, hist_size=5, seed=11)
                host = batch.draw_minibatches(1, 150, seed=4)[0]
                for s in range(21):
                    eng.step(host) if s == 9 else eng.step()
            finally:
                E._StochEngine.AHEAD = old
            runs.append((ahead, eng.z.clone(), eng.psnr_trace()))
    for k in (0, 1):
        for other in (2 + k, 4 + k):
            assert torch.equal(runs[k][1], runs[other][1]) and np.array_equal(runs[k][2], runs[other][2])
