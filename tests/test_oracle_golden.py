"""Pin the oracle (oracle/*.py) to golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import os
import numpy as np
import pytest
from conftest import GOLDEN, golden

import oracle
from oracle import denoise as od, problems as op, loops as ol

IMG256 = os.path.join(GOLDEN, 'synth256.png')
IMG64 = os.path.join(GOLDEN, 'synth64.png')


def csmri(path, n, seed=0, **kw):
    np.random.seed(seed)
    return op.CSMRI(path, H=n, W=n, sample_prob=0.2, snr=20., **kw)


@pytest.mark.parametrize('tag,path,n,mbs', [('s256', IMG256, 256, 1000), ('s64', IMG64, 64, 200)])
def test_csmri_setup_and_grads(g_csmri, tag, path, n, mbs):
    g = g_csmri
    p = csmri(path, n)
    assert np.array_equal(p.mask, g[f'{tag}_mask'])
    assert p.M0 == int(g[f'{tag}_M0'])
    assert np.array_equal(p.Xrec, g[f'{tag}_Xrec'])
    assert abs(p.sigma - float(g[f'{tag}_sigma'])) <= 1e-12 * float(g[f'{tag}_sigma'])
    np.testing.assert_allclose(p.Y, g[f'{tag}_Y'], rtol=0, atol=1e-9)     # dense-DFT-matrix products
    np.testing.assert_allclose(p.Xinit, g[f'{tag}_Xinit'], rtol=0, atol=1e-12)
    np.random.seed(7)
    mb = p.select_mb(mbs)
    assert np.array_equal(mb, g[f'{tag}_mb'])
    np.testing.assert_allclose(p.grad_full(p.Xinit), g[f'{tag}_grad_full'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(p.grad_stoch(p.Xinit, mb), g[f'{tag}_grad_stoch'], rtol=0, atol=1e-12)
    assert p.PSNR(p.Xinit) == float(g[f'{tag}_psnr_init'])
    np.testing.assert_allclose(p.f(p.Xinit), float(g[f'{tag}_f']), rtol=1e-12)
    # the Y terms cancel in a difference of stochastic gradients
    z2 = p.Xinit + 0.01 * np.cos(np.arange(p.N))
    lhs = g[f'{tag}_grad_stoch_z2'] - g[f'{tag}_grad_stoch']
    sel = p.mask * mb
    rhs = np.real(np.fft.ifft2(sel * np.fft.fft2((z2 - p.Xinit).reshape(n, n)))).ravel()
    np.testing.assert_allclose(lhs, rhs, rtol=0, atol=1e-12)


def test_csmri_real_image(g_csmri):
    g = g_csmri
    p = csmri(None, 256, img=g['r256_img'])
    assert np.array_equal(p.mask, g['r256_mask']) and p.M0 == int(g['r256_M0'])
    np.testing.assert_allclose(p.Xinit, g['r256_Xinit'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(p.grad_full(p.Xinit), g['r256_grad_full'], rtol=0, atol=1e-15)


@pytest.mark.parametrize('tag', ['s256', 's64', 'r256'])
def test_sigma_est_and_haar_bayes(g_denoise, tag):
    g = g_denoise
    z0 = g[f'{tag}_z0']
    np.testing.assert_allclose(od.sigma_cols(z0), g[f'{tag}_sigma_cols'], rtol=1e-14)
    s = od.estimate_sigma(z0)
    assert abs(s - float(g[f'{tag}_sigma_est'])) < 1e-15
    np.testing.assert_allclose(od.TVDenoiser().denoise(noisy=z0, sigma_est=s), g[f'{tag}_tv'], rtol=0, atol=1e-14)
    d = od.TVDenoiser(denoise_strength=0.07, decay=0.9)
    np.testing.assert_allclose(d.denoise(noisy=z0, sigma_est=0), g[f'{tag}_tv_strength'], rtol=0, atol=1e-14)
    assert d.t == 1
    np.testing.assert_allclose(od.TVDenoiser(sigma_modifier=1.7).denoise(noisy=z0, sigma_est=s),
                               g[f'{tag}_tv_mod'], rtol=0, atol=1e-14)


def test_sigma_est_edges(g_denoise):
    g = g_denoise
    assert od.estimate_sigma(np.full((64, 64), 0.25)) == pytest.approx(float(g['const_sigma_est']), abs=1e-15)
    out = od.haar_bayes_cols(g['edge_tv_sigma0_in'], 0.0)
    ref = g['edge_tv_sigma0']
    assert np.array_equal(np.isnan(out), np.isnan(ref))            # 0/0 soft-threshold NaNs, as pywt
    np.testing.assert_allclose(out[~np.isnan(ref)], ref[~np.isnan(ref)], atol=1e-14)


def test_fast_exp_known():
    assert od.fast_exp(0.0) == pytest.approx(0.97100782, abs=1e-8)
    y = -np.linspace(0, 30, 1000)
    assert np.all(np.abs(od.fast_exp(y) / np.exp(y) - 1) < 0.07)


@pytest.mark.parametrize('key,kw', [
    ('s64_nlm_h05', dict(h=0.05, sigma=0.05)),
    ('s64_nlm_h005', dict(h=0.005, sigma=0.005)),
])
def test_nlm(g_denoise, key, kw):
    g = g_denoise
    out = od.nl_means_2d(g['s64_z0'], **kw)
    np.testing.assert_array_equal(out, g[key])                    # bit-exact vs the compiled skimage kernel


def test_nlm_denoiser_branches(g_denoise):
    g = g_denoise
    z0, s = g['s64_z0'], float(g['s64_sigma_est'])
    d = od.NLMDenoiser()
    with pytest.raises(AttributeError):
        d.denoise(noisy=z0, sigma_est=s)
    d = od.NLMDenoiser()
    d.sigma = 1.0
    np.testing.assert_array_equal(d.denoise(noisy=z0, sigma_est=s), g['s64_nlm'])
    d = od.NLMDenoiser(denoise_strength=0.1, decay=0.9)
    d.sigma = 0.0
    np.testing.assert_array_equal(d.denoise(noisy=z0, sigma_est=s), g['s64_nlm_strength'])
    np.testing.assert_array_equal(od.nl_means_2d(g['r64_crop'], 0.08, 0.08), g['r64_nlm'])


def test_psnr(g_psnr):
    g = g_psnr
    for w, v, raw in zip(g['ws'], g['psnr'], g['psnr_raw']):
        assert od.psnr_raw(g['Xrec'], w) == pytest.approx(raw, rel=1e-13)
        assert od.psnr(g['Xrec'], w) == v


def _run(name, p, d):
    c = ol.CountingClock
    return {
        'gd': lambda: ol.pnp_gd(p, d, 5e2, 61, converge_check=False, clock=c()),
        'sgd': lambda: ol.pnp_sgd(p, d, 5e2, 51, 200, converge_check=False, lr_decay=0.95, clock=c()),
        'svrg': lambda: ol.pnp_svrg(p, d, 5e2, 60, 4, 200, converge_check=False, clock=c()),
        'saga': lambda: ol.pnp_saga(p, d, 5e2, 53, 200, hist_size=5, converge_check=False, clock=c()),
        'sarah': lambda: ol.pnp_sarah(p, d, 5e2, 70, 4, 200, converge_check=False, lr_decay=0.9, clock=c()),
        'gd_conv': lambda: ol.pnp_gd(p, d, 5e2, 2000, converge_check=True, clock=c()),
        'svrg_conv': lambda: ol.pnp_svrg(p, d, 5e2, 5000, 4, 200, converge_check=True, diverge_check=True, clock=c()),
    }[name]()


@pytest.mark.parametrize('name', ['gd', 'sgd', 'svrg', 'saga', 'sarah', 'gd_conv', 'svrg_conv'])
def test_traces64(g_traces64, name):
    g = g_traces64
    p = csmri(IMG64, 64)
    np.random.seed(1)
    r = _run(name, p, od.TVDenoiser())
    assert list(r['psnr_per_iter']) == list(g[f'{name}_psnr'])
    assert list(r['time_per_iter']) == list(g[f'{name}_time'])
    assert [r['gradient_time'], r['denoise_time']] == list(g[f'{name}_gt_dt'])
    np.testing.assert_allclose(r['z'], g[f'{name}_z'], rtol=0, atol=1e-12)


def test_true_svrg_64(g_traces64):
    g = g_traces64
    p = csmri(IMG64, 64)
    np.random.seed(1)
    # 3 outer x 4 inner: clock budget 2 + 3*(3 + 5*4) = 71 ticks
    r = ol.pnp_svrg(p, od.TVDenoiser(), 5e2, 2 + 3 * 23 - 1, 4, 200, converge_check=False,
                    clock=ol.CountingClock(), variant='svrg')
    assert list(r['psnr_per_iter']) == list(g['truesvrg_psnr'])
    np.testing.assert_allclose(r['z'], g['truesvrg_z'], rtol=0, atol=1e-12)


def test_traces256(g_traces256):
    g = g_traces256
    for variant, key in (('reference', 'svrg'), ('svrg', 'truesvrg')):
        p = csmri(IMG256, 256)
        np.random.seed(1)
        r = ol.pnp_svrg(p, od.TVDenoiser(), 2e3, 2 + 4 * 53 - (1 if variant == 'svrg' else 0), 10, 1000,
                        converge_check=False, clock=ol.CountingClock(), variant=variant)
        assert list(r['psnr_per_iter']) == list(g[f'{key}_psnr'])
        np.testing.assert_allclose(r['z'], g[f'{key}_z'], rtol=0, atol=1e-12)


@pytest.mark.parametrize('variant,key', [('reference', 'svrg'), ('svrg', 'truesvrg')])
def test_traces256_full_length(g_traces256_full, variant, key):
    """SURVEY 8(d) at its full length -- 20 outer x 10 inner iterations, eta 2e3, mb 1000, TV prox -- against the trace the
    real reference produced (tests/golden/make_golden_r3.py): the oracle that the GPU tests of the same length lean on."""
    g = g_traces256_full
    p = csmri(IMG256, 256)
    np.random.seed(1)
    r = ol.pnp_svrg(p, od.TVDenoiser(), 2e3, 2 + 20 * 53 - (1 if variant == 'svrg' else 0), 10, 1000,
                    converge_check=False, clock=ol.CountingClock(), variant=variant)
    assert len(r['psnr_per_iter']) == 221 and list(r['psnr_per_iter']) == list(g[f'{key}_psnr'])
    np.testing.assert_allclose(r['z'], g[f'{key}_z'], rtol=0, atol=1e-11)


def test_deblur(g_deblur):
    g = g_deblur
    np.random.seed(0)
    p = op.Deblur(IMG256, H=256, W=256, kernel='Minimal', scale_percent=100, snr=5.)
    assert p.sigma == pytest.approx(float(g['min256_sigma']), rel=1e-13)
    assert np.flatnonzero(p.B).tolist() == [0, 32832, 32853, 32896]
    np.testing.assert_allclose(p.Y, g['min256_Y'], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(p.Xinit, g['min256_Xinit'])
    np.random.seed(3)
    mb = p.select_mb(3000)
    assert np.array_equal(mb, g['min256_mb'])
    np.testing.assert_allclose(p.grad_full(p.Xinit), g['min256_grad_full'], rtol=0, atol=1e-18)
    np.testing.assert_allclose(p.grad_stoch(p.Xinit, mb), g['min256_grad_stoch'], rtol=0, atol=1e-14)
    # the reference's one deterministic published number (deblur notebook cell 4), via the 01.png fixture
    assert float(g['known_sigma_01png']) == pytest.approx(0.0015155036596592854, rel=1e-14)
    assert int(g['known_M_01png']) == 65536


def test_deblur_adjoint_identity():
    # FFT(roll(flip(B),1)) == conj(FFT(B))
    B = np.random.default_rng(0).random(4096)
    np.testing.assert_allclose(np.fft.fft(np.roll(np.flip(B), 1)), np.conj(np.fft.fft(B)), atol=1e-10)


def test_bilinear_adjoint_dot():
    """pylops Bilinear is restated from its semantics only ("parity unpinned"): dot-test."""
    np.random.seed(0)
    p = op.Deblur(IMG64, H=64, W=64, kernel='Minimal', scale_percent=50, snr=20.)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(p.N), rng.standard_normal(p.M)
    assert p.M == 32 * 32
    assert np.dot(p.Bop.matvec(x), y) == pytest.approx(np.dot(x, p.Bop.rmatvec(y)), rel=1e-12)


def test_pr(g_pr):
    g = g_pr
    np.random.seed(0)
    p = op.PhaseRetrieval(IMG64, H=32, W=32, num_meas=5 * 1024, snr=20.)
    np.testing.assert_allclose([p.A.sum(), np.abs(p.A).sum()], g['pr_A_checksum'], rtol=1e-13)
    np.testing.assert_allclose(p.Y, g['pr_Y'], rtol=1e-12)
    np.testing.assert_allclose(p.Xinit, g['pr_Xinit'], rtol=0, atol=1e-9)
    np.random.seed(3)
    mb = p.select_mb(700)
    assert np.array_equal(mb, g['pr_mb'])
    np.testing.assert_allclose(p.grad_full(g['pr_Xinit']), g['pr_grad_full'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(p.grad_stoch(g['pr_Xinit'], mb), g['pr_grad_stoch'], rtol=1e-9, atol=1e-10)


def test_saga_nlm_deblur(g_deblur):
    g = g_deblur
    np.random.seed(0)
    # kernel.png is reference data: rebuild B from the fixture instead of the file
    p = op.Deblur(IMG64, H=64, W=64, kernel=g['k64_B'] * 4096, scale_percent=100, snr=20.)
    assert p.sigma == pytest.approx(float(g['k64_sigma']), rel=1e-13)
    np.testing.assert_allclose(p.Y, g['k64_Y'], rtol=1e-13)
    np.random.seed(3)
    mb = p.select_mb(500)
    np.testing.assert_allclose(p.grad_full(p.Xinit), g['k64_grad_full'], rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(p.grad_stoch(p.Xinit, mb), g['k64_grad_stoch'], rtol=1e-11, atol=1e-12)
    np.random.seed(1)
    d = od.NLMDenoiser()
    d.sigma = 1.0
    r = ol.pnp_saga(p, d, 1.0, 33, 500, hist_size=4, converge_check=False, clock=ol.CountingClock())
    assert list(r['psnr_per_iter']) == list(g['k64_saga_nlm_psnr'])
    np.testing.assert_allclose(r['z'], g['k64_saga_nlm_z'], rtol=0, atol=1e-10)


def test_dncnn_oracle():
    from conftest import golden
    w = dict(golden('dncnn_noise15.npz'))
    io = golden('dncnn_io.npz')
    r = od.dncnn_forward(w, io['net64_in'])
    np.testing.assert_allclose(r, io['net64_out'], rtol=0, atol=2e-6)
    den = golden('denoise.npz')
    out = od.DnCNNDenoiser(w, 15).denoise(noisy=den['s64_z0'], sigma_est=0.1)
    np.testing.assert_allclose(out, io['den64_s15'], rtol=0, atol=5e-6)


def test_mmo_oracle():
    """oracle.MMODenoiser (transpose, clamp, bias/LeakyReLU/skip net) == the reference's MMODenoiser.denoise run on its
    own simple_CNN class (tests/golden/make_golden_mmo.py), square and rectangular images."""
    g = golden('mmo_seeded.npz')
    w = {k: g[k] for k in g.files if k.startswith('conv') or k in ('n_layers', 'negative_slope')}
    den = od.MMODenoiser(w)
    for name in ('sq', 'rect'):
        y = den.denoise(g[f'{name}_in'])
        assert y.dtype == np.float32 and y.shape == g[f'{name}_out'].shape
        assert np.abs(y - g[f'{name}_out']).max() <= 2e-6
    assert den.t == 2


# ---------------------------------------------------------------------------------- round-2 fixtures (make_golden_r2.py)
def _png(tmp_path, name, pixels):
    from PIL import Image
    path = str(tmp_path / name)
    Image.fromarray(pixels).save(path)
    return path


def test_resize_and_kernel_png(g_r2, tmp_path):
    """a6: PIL's bicubic 512^2 -> 256^2 resize + min-max of the reference's data/Set12/08.png pixels; a13: the
    kernel_path branch (kernel.png resized to H x W, / N) at 64^2 and 256^2."""
    g = g_r2
    src = _png(tmp_path, '08.png', g['resize_pixels'])
    p = op.Problem(src, 256, 256)
    assert np.array_equal(p.Xrec, g['resize_Xrec'])
    kp = _png(tmp_path, 'kernel.png', g['kernelpng_pixels'])
    np.random.seed(0)
    p64 = op.Deblur(IMG64, H=64, W=64, kernel_path=kp, scale_percent=100, snr=20.)
    assert np.array_equal(p64.B, g['kernelpng_B64'])
    np.random.seed(0)
    p = op.Deblur(IMG256, H=256, W=256, kernel_path=kp, scale_percent=100, snr=20.)
    assert np.array_equal(p.B, g['kernelpng_B256'])
    assert p.sigma == pytest.approx(float(g['kernelpng_sigma256']), rel=1e-13)
    np.testing.assert_allclose(p.Y, g['kernelpng_Y256'], rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(p.Xinit, g['kernelpng_Xinit256'])
    gf = g['kernelpng_grad_full256']
    assert np.abs(p.grad_full(p.Xinit) - gf).max() <= 1e-11 * np.abs(gf).max()


def test_nlm_256(g_r2, g_denoise):
    """a19 at BASELINE's size: NLMDenoiser.denoise on a 256 x 256 iterate, bit-exact."""
    d = od.NLMDenoiser()
    d.sigma = 1.0
    s = od.estimate_sigma(g_denoise['r256_z0'])
    assert s == float(g_r2['nlm256_sigma_est'])
    assert np.array_equal(d.denoise(noisy=g_denoise['r256_z0'], sigma_est=s), g_r2['nlm256_out'])


def test_config4_full_size(g_r2):
    """BASELINE config 4 at 256 x 256: Deblur ("Minimal") gradients and the pnp_saga + NLM trace of the reference."""
    g = g_r2
    np.random.seed(0)
    p = op.Deblur(IMG256, H=256, W=256, kernel='Minimal', scale_percent=100, snr=20.)
    assert p.sigma == pytest.approx(float(g['c4_sigma']), rel=1e-13)
    np.testing.assert_allclose(p.Y, g['c4_Y'], rtol=0, atol=1e-15)
    np.random.seed(3)
    mb = p.select_mb(3000)
    assert np.array_equal(mb, g['c4_mb'])
    for got, key in ((p.grad_full(p.Xinit), 'c4_grad_full'), (p.grad_stoch(p.Xinit, mb), 'c4_grad_stoch')):
        assert np.abs(got - g[key]).max() <= 1e-11 * np.abs(g[key]).max()
    np.random.seed(1)
    d = od.NLMDenoiser()
    d.sigma = 1.0
    r = ol.pnp_saga(p, d, 1e9, 5 * 6 - 1, 3000, hist_size=4, converge_check=False, clock=ol.CountingClock())
    assert list(r['psnr_per_iter']) == list(g['c4_saga_nlm_psnr'])
    np.testing.assert_allclose(r['z'], g['c4_saga_nlm_z'], rtol=0, atol=1e-9)


@pytest.mark.parametrize('sigma', [5, 40])
def test_dncnn_other_noise_levels(sigma, g_denoise):
    """a20 at the other noise levels the reference ships weights for: wrapper outputs of the reference class."""
    g = golden(f'dncnn_noise{sigma}.npz')
    w = {k: g[k] for k in g.files if k.startswith(('conv', 'bn')) or k == 'n_layers'}
    d = od.DnCNNDenoiser(w, sigma)
    np.testing.assert_allclose(d.denoise(noisy=g_denoise['s64_z0']), g['den64'], rtol=0, atol=5e-6)
    np.testing.assert_allclose(d.denoise(noisy=g_denoise['r256_z0']), g['den256'], rtol=0, atol=5e-6)
