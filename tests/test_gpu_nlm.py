"""GPU parity of the NLM prox against the compiled skimage kernel's outputs (golden, produced by
the reference's NLMDenoiser / denoise_nl_means) and the oracle.  f64: BIT-EXACT (integer fast_exp
trick + the reference's accumulation order); f32: 2e-4 absolute (fast_exp is a step function of
the distance, so single-precision distances can move a weight by one fast_exp quantum)."""
import numpy as np
import pytest
import torch

from oracle import denoise as od

pytestmark = pytest.mark.gpu


def dev(x, dtype):
    return torch.from_numpy(np.ascontiguousarray(x)).to('cuda', dtype)


@pytest.mark.parametrize('key,h', [('s64_nlm_h05', 0.05), ('s64_nlm_h005', 0.005)])
def test_nlm_bit_exact_f64(g_denoise, key, h):
    from pnp_svrg_amd import ops
    z = g_denoise['s64_z0']
    sig = torch.full((1,), h, dtype=torch.float64, device='cuda')
    out, _ = ops.nlm2d(dev(z[None], torch.float64), sigma_in=sig)
    np.testing.assert_array_equal(out[0].cpu().numpy(), g_denoise[key])


def test_nlm_real_crop_and_batch(g_denoise):
    from pnp_svrg_amd import ops
    crop = g_denoise['r64_crop']
    zb = np.stack([crop, g_denoise['s64_z0']])
    sig = torch.tensor([0.08, 0.05], dtype=torch.float64, device='cuda')
    xrec = np.clip(zb, 0, 1)
    out, sse = ops.nlm2d(dev(zb, torch.float64), sigma_in=sig, xrec=dev(xrec, torch.float64))
    o = out.cpu().numpy()
    np.testing.assert_array_equal(o[0], g_denoise['r64_nlm'])
    np.testing.assert_array_equal(o[1], g_denoise['s64_nlm_h05'])
    np.testing.assert_allclose(sse.cpu().numpy(), ((xrec - o) ** 2).reshape(2, -1).sum(1), rtol=1e-12)


def test_nlm_f32(g_denoise):
    from pnp_svrg_amd import ops
    z = g_denoise['s64_z0']
    sig = torch.full((1,), 0.05, dtype=torch.float32, device='cuda')
    out, _ = ops.nlm2d(dev(z[None], torch.float32), sigma_in=sig)
    assert np.abs(out[0].cpu().numpy() - g_denoise['s64_nlm_h05']).max() <= 2e-4


def test_nlm_non_multiple_of_tile_and_other_params():
    """40 x 56 image (ragged 16 x 16 tiles), patch 7, distance 3, against the oracle."""
    from pnp_svrg_amd import ops
    rng = np.random.default_rng(0)
    z = rng.random((40, 56))
    ref = od.nl_means_2d(z, 0.1, 0.1, patch_size=7, patch_distance=3)
    sig = torch.full((1,), 0.1, dtype=torch.float64, device='cuda')
    out, _ = ops.nlm2d(dev(z[None], torch.float64), sigma_in=sig, patch_size=7, patch_distance=3)
    np.testing.assert_array_equal(out[0].cpu().numpy(), ref)
    ref = od.nl_means_2d(z, 0.2, 0.0, patch_size=3, patch_distance=8)          # fixed-h branch, var = 0
    out, _ = ops.nlm2d(dev(z[None], torch.float64), fixed_h=0.2, patch_size=3, patch_distance=8)
    np.testing.assert_array_equal(out[0].cpu().numpy(), ref)


def test_nlm_strip_kernel_equals_lds_kernel():
    """The reference's configuration (patch 5, distance 5) runs on the register-strip kernel -- in f64 for batches of up to
    4 images, the LDS-streaming kernel beyond: same arithmetic in the same order, so a ragged 40 x 56 image gives the same
    bits alone (strip), in a batch of 6 (LDS form) and from the oracle; in f32 (strip at every batch size) a batched image
    equals its single-image result."""
    from pnp_svrg_amd import ops
    rng = np.random.default_rng(3)
    zb = rng.random((6, 40, 56))
    sig = np.full(6, 0.07)
    ref = od.nl_means_2d(zb[2], 0.07, 0.07)
    one, _ = ops.nlm2d(dev(zb[2:3], torch.float64), sigma_in=dev(sig[:1], torch.float64))
    six, _ = ops.nlm2d(dev(zb, torch.float64), sigma_in=dev(sig, torch.float64))
    np.testing.assert_array_equal(one[0].cpu().numpy(), ref)
    np.testing.assert_array_equal(six[2].cpu().numpy(), ref)
    one32, _ = ops.nlm2d(dev(zb[2:3], torch.float32), sigma_in=dev(sig[:1], torch.float32))
    six32, _ = ops.nlm2d(dev(zb, torch.float32), sigma_in=dev(sig, torch.float32))
    assert torch.equal(one32[0], six32[2])
    assert np.abs(one32[0].cpu().numpy() - ref).max() <= 2e-4


def test_nlm_denoiser_surface(g_denoise):
    import denoisers
    g = g_denoise
    z0, s = g['s64_z0'], float(g['s64_sigma_est'])
    d = denoisers.NLMDenoiser(dtype=torch.float64)
    with pytest.raises(AttributeError):                          # reads self.sigma, never set
        d.denoise(noisy=z0, sigma_est=s)
    d = denoisers.NLMDenoiser(dtype=torch.float64)
    d.sigma = 1.0
    np.testing.assert_array_equal(d.denoise(noisy=z0, sigma_est=s), g['s64_nlm'])
    assert d.t == 1
    d = denoisers.NLMDenoiser(denoise_strength=0.1, decay=0.9, dtype=torch.float64)
    d.sigma = 0.0
    np.testing.assert_array_equal(d.denoise(noisy=z0, sigma_est=s), g['s64_nlm_strength'])
