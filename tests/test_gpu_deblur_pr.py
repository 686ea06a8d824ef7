"""GPU parity of the Deblur (1-D FFT blur) and phase-retrieval gradients, and of the config-4 style
loop (pnp_saga + NLM on Deblur), against golden vectors from the reference and the oracle."""
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN

from oracle import problems as op, loops as ol, denoise as od

pytestmark = pytest.mark.gpu
IMG256 = os.path.join(GOLDEN, 'synth256.png')
IMG64 = os.path.join(GOLDEN, 'synth64.png')


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_deblur_minimal_256(g_deblur, dtype):
    import problems
    g = g_deblur
    np.random.seed(0)
    p = problems.Deblur(IMG256, H=256, W=256, kernel='Minimal', scale_percent=100, snr=5., dtype=dtype)
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    assert p.sigma == pytest.approx(float(g['min256_sigma']), rel=tol)
    assert p.M == 65536 and p.lrH == 256
    np.testing.assert_array_equal(p.Xinit, g['min256_Xinit'])          # RNG stream position preserved
    if dtype == torch.float64:
        np.testing.assert_allclose(p.Y, g['min256_Y'], rtol=0, atol=1e-13)
    np.random.seed(3)
    mb = p.select_mb(3000)
    assert np.array_equal(mb, g['min256_mb'])
    gf, gs = p.grad_full(g['min256_Xinit']), p.grad_stoch(g['min256_Xinit'], mb)
    sf, ss = np.abs(g['min256_grad_full']).max(), np.abs(g['min256_grad_stoch']).max()
    assert np.abs(gf - g['min256_grad_full']).max() <= (1e-11 if dtype == torch.float64 else 2e-4) * sf
    assert np.abs(gs - g['min256_grad_stoch']).max() <= (1e-11 if dtype == torch.float64 else 2e-4) * ss


def _k64(problems, g, dtype):
    np.random.seed(0)
    return problems.Deblur(IMG64, H=64, W=64, kernel=g['k64_B'] * 4096, scale_percent=100, snr=20., dtype=dtype)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_deblur_kernel_image_64(g_deblur, dtype):
    import problems
    g = g_deblur
    p = _k64(problems, g, dtype)
    np.random.seed(3)
    mb = p.select_mb(500)
    rel = 1e-10 if dtype == torch.float64 else 2e-4
    for got, key in ((p.grad_full(p.Xinit), 'k64_grad_full'), (p.grad_stoch(p.Xinit, mb), 'k64_grad_stoch')):
        assert np.abs(got - g[key]).max() <= rel * np.abs(g[key]).max()
    np.testing.assert_allclose(p.f(p.Xinit), op.Deblur.f(_OracleView(p), p.Xinit), rtol=1e-6)


class _OracleView:
    """minimal view so the oracle's f() formula can be evaluated on the product's data"""
    def __init__(self, p):
        self.Y, self.M, self.N, self.B, self.Bop = p.Y, p.M, p.N, p.B, None
    _down = op.Deblur._down
    fft_blur = op.Deblur.fft_blur
    forward_model = op.Deblur.forward_model


def test_saga_nlm_deblur_loop(g_deblur):
    """BASELINE config 4 shape (small): pnp_saga + NLM prox on Deblur, f64 device path vs the
    reference's golden trace (identical RNG stream: select_mb then np.random.choice(hist, 1))."""
    import algorithms, problems, denoisers
    g = g_deblur
    p = _k64(problems, g, torch.float64)
    np.random.seed(1)
    d = denoisers.NLMDenoiser()
    d.sigma = 1.0
    r = algorithms.pnp_saga(p, d, 1.0, 33, 500, hist_size=4, verbose=False, converge_check=False, clock=ol.CountingClock())
    assert list(r['psnr_per_iter']) == list(g['k64_saga_nlm_psnr'])
    np.testing.assert_allclose(r['z'], g['k64_saga_nlm_z'], rtol=0, atol=1e-8)


def test_deblur_bilinear_adjoint_and_oracle():
    """scale_percent = 50: pylops Bilinear is restated (parity unpinned) -> adjoint dot-test on the
    device operator, plus agreement with the oracle's restatement."""
    import problems
    np.random.seed(0)
    p = problems.Deblur(IMG64, H=64, W=64, kernel='Minimal', scale_percent=50, snr=20., dtype=torch.float64)
    np.random.seed(0)
    po = op.Deblur(IMG64, H=64, W=64, kernel='Minimal', scale_percent=50, snr=20.)
    assert p.M == 1024 and (p.lrH, p.lrW) == (32, 32)
    np.testing.assert_allclose(p.Y, po.Y, rtol=0, atol=1e-12)
    np.random.seed(3)
    mb = p.select_mb(300)
    np.testing.assert_allclose(p.grad_full(po.Xinit), po.grad_full(po.Xinit), rtol=0, atol=1e-14)
    np.testing.assert_allclose(p.grad_stoch(po.Xinit, mb), po.grad_stoch(po.Xinit, mb), rtol=0, atol=1e-11)
    # <S B x, y> == <x, B^T S^T y>: grad with Y = 0 and all-ones selector is B^T S^T S B x
    rng = np.random.default_rng(1)
    x = rng.standard_normal(p.N)
    SBx = p.forward_model(x)
    zero = torch.zeros(1, p.M, dtype=torch.float64, device='cuda')
    AtAx = p.plan.grad(p.to_device(x).reshape(1, -1), zero).double().cpu().numpy().ravel()
    assert np.dot(SBx, SBx) == pytest.approx(np.dot(x, AtAx), rel=1e-11)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_phase_retrieval(g_pr, dtype):
    import algorithms, problems, denoisers
    g = g_pr
    np.random.seed(0)
    p = problems.PhaseRetrieval(IMG64, H=32, W=32, num_meas=5 * 1024, snr=20., dtype=dtype)
    np.testing.assert_allclose(p.Y, g['pr_Y'], rtol=1e-12)
    np.testing.assert_allclose(p.Xinit, g['pr_Xinit'], rtol=0, atol=1e-9)
    np.random.seed(3)
    mb = p.select_mb(700)
    assert np.array_equal(mb, g['pr_mb'])
    rel = 1e-9 if dtype == torch.float64 else 5e-4
    for got, key in ((p.grad_full(g['pr_Xinit']), 'pr_grad_full'), (p.grad_stoch(g['pr_Xinit'], mb), 'pr_grad_stoch')):
        assert np.abs(got - g[key]).max() <= rel * np.abs(g[key]).max()
    np.random.seed(1)
    r = algorithms.pnp_svrg(p, denoisers.TVDenoiser(), 0.2, 40, 4, 700, verbose=False, converge_check=False,
                            clock=ol.CountingClock())
    ps = np.array(r['psnr_per_iter'])
    assert len(ps) == len(g['pr_svrg_psnr'])
    if dtype == torch.float64:
        assert list(ps) == list(g['pr_svrg_psnr'])
        np.testing.assert_allclose(r['z'], g['pr_svrg_z'], rtol=0, atol=1e-8)
    else:
        assert np.abs(ps - g['pr_svrg_psnr']).max() <= 0.01 + 1e-9


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('scale', [100, 50])
def test_deblur_128_vs_oracle(dtype, scale):
    """128 x 128 (rectangular 8 x 16 FFT split, N = 16384): gradients against the golden-pinned oracle."""
    import problems
    np.random.seed(0)
    p = problems.Deblur(IMG256, H=128, W=128, kernel='Minimal', scale_percent=scale, snr=10., dtype=dtype)
    np.random.seed(0)
    po = op.Deblur(IMG256, H=128, W=128, kernel='Minimal', scale_percent=scale, snr=10.)
    assert p.M == po.M
    np.random.seed(3)
    mb = p.select_mb(p.M // 5)
    rel = 1e-10 if dtype == torch.float64 else 3e-4
    for got, ref in ((p.grad_full(po.Xinit), po.grad_full(po.Xinit)), (p.grad_stoch(po.Xinit, mb), po.grad_stoch(po.Xinit, mb))):
        assert np.abs(got - ref).max() <= rel * np.abs(ref).max()


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-12), (torch.float32, 2e-5)])
def test_pr_spectral_apply(dtype, tol):
    """pnp_pr_spectral_apply == D v for D = A^T diag(y) A / M (reference PR.py:53,59), ragged M and N."""
    from pnp_svrg_amd import ops
    rng = np.random.default_rng(3)
    M, N = 301, 250
    A, y, v = rng.standard_normal((M, N)), np.abs(rng.standard_normal(M)), rng.standard_normal(N)
    ref = (A.T @ (A * y[:, None]) / M) @ v
    dev = lambda a: torch.from_numpy(a).to('cuda', dtype)
    got = ops.pr_spectral_apply(dev(A), dev(v), dev(y), scale=1.0 / M).double().cpu().numpy()
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()


def test_pr_spectral_init_matches_matrix_power_iteration():
    """Device power iteration (never forms D) vs the oracle's restatement of PR.py:50-63 (forms D) on a problem that
    is not the golden one: same Xinit to 1e-9, in both storage dtypes of the problem."""
    from oracle import problems as op
    from pnp_svrg_amd.problems import PhaseRetrieval
    import os
    img = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'synth64.png')
    np.random.seed(5)
    ref = op.PhaseRetrieval(img, H=24, W=24, num_meas=3 * 576, snr=25.)
    for dt in (torch.float64, torch.float32):
        np.random.seed(5)
        p = PhaseRetrieval(img, H=24, W=24, num_meas=3 * 576, snr=25., dtype=dt)
        np.testing.assert_allclose(p.Xinit, ref.Xinit, rtol=0, atol=1e-9)
        assert p._A_d.dtype == dt and not hasattr(p, '_A64_d')
