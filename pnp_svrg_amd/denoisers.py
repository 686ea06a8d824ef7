"""Host-side mirror of the reference's `denoisers/*` interface, backed by the HIP kernels.

Same class names, constructor arguments, `t` counter and `denoise(noisy=<HxW>, sigma_est=<float>)`
protocol as reference denoisers/denoiser.py, TV.py, NLM.py, RealSN_DnCNN.py, BM3D.py.
`denoise` takes/returns NumPy float64 like the reference; `denoise_device` is the form the
loops use (device tensor in, device tensor out, optional fused noise estimate and PSNR error).
"""
import numpy as np
import torch

from . import ops


def _as_dev(noisy, dtype=None):
    if isinstance(noisy, torch.Tensor):
        return noisy
    from .problems import get_default_dtype
    return torch.from_numpy(np.ascontiguousarray(noisy, dtype=np.float64)).to('cuda', dtype or get_default_dtype())


class Denoise():
    """reference denoisers/denoiser.py:2-7."""

    def __init__(self):
        self.t = 0

    def denoise(self, noisy):
        raise NotImplementedError('Need to implement denoise() method')


class TVDenoiser(Denoise):
    """reference denoisers/TV.py:9-26: despite the name, skimage's wavelet BayesShrink applied
    to every column of the 2-D image.  Runs as pnp_prox_tv on the MI355X."""

    def __init__(self, multi=True, rescale_sigma=True, decay=1, denoise_strength=0, sigma_modifier=1, dtype=None):
        super().__init__()
        if not multi:
            raise NotImplementedError('multi=False (a true 2-D wavelet transform) is not on the reference hot path')
        self.multi = multi
        self.rescale_sigma = rescale_sigma
        self.denoise_strength = denoise_strength
        self.sigma_modifier = sigma_modifier
        self.decay = decay
        self.dtype = dtype

    # loops call this: sigma_est=None -> estimated inside the same kernel (estimate_sigma fused)
    def denoise_device(self, z, sigma_est=None, xrec=None):
        """z: [B,H,W] device tensor.  Returns (denoised, sse or None, sigma_est [B])."""
        self.t += 1
        fallback = self.denoise_strength * self.decay ** self.t
        return ops.prox_tv(z, sigma_in=sigma_est, sigma_modifier=self.sigma_modifier, fallback_sigma=fallback, xrec=xrec)

    def denoise(self, noisy, sigma_est=0):
        z = _as_dev(noisy, self.dtype)
        H, W = z.shape[-2:]
        s = torch.full((1,), float(sigma_est), dtype=z.dtype, device=z.device)
        out, _, _ = self.denoise_device(z.reshape(1, H, W), sigma_est=s)
        if isinstance(noisy, torch.Tensor):
            return out.reshape(noisy.shape)
        return out.reshape(H, W).double().cpu().numpy()


class BM3DDenoiser(Denoise):
    """reference denoisers/BM3D.py wraps the closed-source PyPI `bm3d` binaries: out of scope
    (SURVEY section 2).  Kept importable so `from denoisers import *` works."""

    def __init__(self, decay=1, denoise_strength=0, sigma_modifier=1):
        super().__init__()
        self.decay, self.denoise_strength, self.sigma_modifier = decay, denoise_strength, sigma_modifier

    def denoise(self, noisy, sigma_est=0):
        raise NotImplementedError('BM3D is a third-party binary plug-in; not part of the MI355X hot path')
