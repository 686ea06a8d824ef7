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
    ops.require_gpu()                      # fail loudly (NativeError) when there is no MI355X / no library
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

    def prox_inplace(self, z, xrec, sse, probe=False):
        """estimate_sigma + denoise + error sum on z in place, nothing allocated, no per-call host state: what a
        hipGraph can hold.  probe=True only answers whether this instance can (the decaying fixed strength cannot)."""
        if probe:
            return self.denoise_strength == 0
        self.t += 1
        ops.prox_tv(z, sigma_modifier=self.sigma_modifier, fallback_sigma=0.0, xrec=xrec, out=z, sse=sse)
        return True


class BM3DDenoiser(Denoise):
    """reference denoisers/BM3D.py wraps the closed-source PyPI `bm3d` binaries: out of scope
    (SURVEY section 2).  Kept importable so `from denoisers import *` works."""

    def __init__(self, decay=1, denoise_strength=0, sigma_modifier=1):
        super().__init__()
        self.decay, self.denoise_strength, self.sigma_modifier = decay, denoise_strength, sigma_modifier

    def denoise(self, noisy, sigma_est=0):
        raise NotImplementedError('BM3D is a third-party binary plug-in; not part of the MI355X hot path')


# ------------------------------------------------------------------------------------------
# DnCNN family (reference denoisers/RealSN_DnCNN.py + DeepDenoisers/utils/utils.py:10-33)
# ------------------------------------------------------------------------------------------
def dncnn_weights_from_state_dict(sd):
    """Reference checkpoint (state dict of tensors) -> {'n_layers', 'conv{i}.weight', 'bn{i}.*'} arrays.

    Handles the `module.` prefix nn.DataParallel left in the files and the RealSN layout: at
    inference the spectral-norm hook just uses the stored `weight` buffer, so
    `weight_orig` / `weight_u` are ignored."""
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
    conv_ids = sorted({int(k.split('.')[1]) for k, v in sd.items() if k.endswith('.weight') and v.dim() == 4})
    out = {'n_layers': np.int64(len(conv_ids))}
    for i, ci in enumerate(conv_ids):
        out[f'conv{i}.weight'] = sd[f'dncnn.{ci}.weight'].detach().cpu().numpy().astype(np.float32)
        bi = ci + 1
        if f'dncnn.{bi}.running_mean' in sd:
            out[f'bn{i}.weight'] = sd[f'dncnn.{bi}.weight'].cpu().numpy()
            out[f'bn{i}.bias'] = sd[f'dncnn.{bi}.bias'].cpu().numpy()
            out[f'bn{i}.mean'] = sd[f'dncnn.{bi}.running_mean'].cpu().numpy()
            out[f'bn{i}.var'] = sd[f'dncnn.{bi}.running_var'].cpu().numpy()
    return out


def random_dncnn_weights(n_layers=17, seed=0):
    """Random-init weights of the DnCNN-17 architecture (for synthetic benchmarks: there is no
    network to fetch checkpoints).  He-style scaling keeps activations O(1) through 17 layers."""
    rng = np.random.default_rng(seed)
    w = {'n_layers': np.int64(n_layers)}
    for i in range(n_layers):
        cin = 1 if i == 0 else 64
        cout = 1 if i == n_layers - 1 else 64
        w[f'conv{i}.weight'] = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
        if 0 < i < n_layers - 1:
            w[f'bn{i}.weight'] = (1.0 + 0.1 * rng.standard_normal(64)).astype(np.float32)
            w[f'bn{i}.bias'] = (0.05 * rng.standard_normal(64)).astype(np.float32)
            w[f'bn{i}.mean'] = (0.1 * rng.standard_normal(64)).astype(np.float32)
            w[f'bn{i}.var'] = (1.0 + 0.2 * rng.random(64)).astype(np.float32)
    return w


def load_model(model_type, sigma):
    """reference DeepDenoisers/utils/utils.py:10-33: same CWD-relative path and model types
    ('DnCNN', 'SimpleCNN', 'RealSN_DnCNN', 'RealSN_SimpleCNN', anything else -> RealSN_DnCNN layout).
    Returns the weight dict the MFMA plan consumes (there is no torch module on this path)."""
    known = ('DnCNN', 'SimpleCNN', 'RealSN_DnCNN', 'RealSN_SimpleCNN')
    path = "./denoisers/DeepDenoisers/Pretrained_models/" + model_type + "_noise" + str(sigma) + ".pth"
    if model_type not in known:
        path = "./denoisers/DeepDenoisers/Pretrained_models/" + str(model_type) + "_noise" + str(sigma) + ".pth"
    sd = torch.load(path, map_location='cpu', weights_only=True)       # FileNotFoundError like the reference
    return dncnn_weights_from_state_dict(sd)


class RealSN_DnCNNDenoiser(Denoise):
    """reference denoisers/RealSN_DnCNN.py:8-42.  `model_type`, `sigma` as in the reference; the
    network runs as the MFMA conv stack of pnp_dncnn_* (fp32, like the reference's net).
    Extension: `weights=` (a weight dict) bypasses the checkpoint file.  Like the reference this
    denoiser ignores `sigma_est` and does not advance `t`."""

    def __init__(self, model_type, sigma, *, weights=None):
        super().__init__()
        self.model_type = model_type
        self.sigma = sigma
        self.model = weights if weights is not None else load_model(self.model_type, self.sigma)
        self._plans = {}

    def _plan(self, B, H, W):
        key = (B, H, W)
        if key not in self._plans:
            self._plans[key] = ops.DncnnPlan(self.model, H, W, B)
        return self._plans[key]

    def denoise_device(self, z, sigma_est=None, xrec=None, out=None, sse=None):
        B, H, W = z.shape
        out, sse = self._plan(B, H, W).denoise(z, self.sigma, xrec=xrec, out=out, sse=sse)
        return out, sse, None

    def prox_inplace(self, z, xrec, sse, probe=False):
        """denoise + error sum on z in place (plan-owned workspaces: nothing allocated), for hipGraph capture."""
        if probe:
            return True
        B, H, W = z.shape
        self._plan(B, H, W).denoise(z, self.sigma, xrec=xrec, out=z, sse=sse)
        return True

    def denoise(self, noisy, sigma_est=0):
        z = _as_dev(noisy)
        H, W = z.shape[-2:]
        out, _, _ = self.denoise_device(z.reshape(1, H, W))
        if isinstance(noisy, torch.Tensor):
            return out.reshape(noisy.shape)
        return out.reshape(H, W).double().cpu().numpy()


class NLMDenoiser(Denoise):
    """reference denoisers/NLM.py:9-27 (skimage slow-mode non-local means) on the MI355X.
    Like the reference, `denoise` reads `self.sigma`, which the constructor does not set:
    callers assign `denoiser.sigma` first, otherwise AttributeError -- kept for drop-in parity."""

    def __init__(self, decay=1, denoise_strength=0, patch_size=4, patch_distance=5, sigma_modifier=1, fast_mode=False,
                 multichannel=True, dtype=None):
        super().__init__()
        if fast_mode:
            raise NotImplementedError('fast_mode=True (integral-image NLM) is not on the reference hot path')
        self.decay = decay
        self.denoise_strength = denoise_strength
        self.fast_mode = fast_mode
        self.sigma_modifier = sigma_modifier
        self.patch = dict(patch_size=patch_size, patch_distance=patch_distance, multichannel=multichannel)
        self.dtype = dtype

    def denoise_device(self, z, sigma_est=None, xrec=None):
        """z [B,H,W]; sigma_est: device tensor [B] or None (estimated on the device first)."""
        self.t += 1
        ps, pd = self.patch['patch_size'], self.patch['patch_distance']
        if self.sigma > 0:
            if sigma_est is None:
                sigma_est = ops.sigma_est(z)
            out, sse = ops.nlm2d(z, sigma_in=sigma_est, sigma_modifier=self.sigma_modifier, patch_size=ps,
                                 patch_distance=pd, xrec=xrec)
        else:
            out, sse = ops.nlm2d(z, fixed_h=self.denoise_strength * self.decay ** self.t, patch_size=ps,
                                 patch_distance=pd, xrec=xrec)
        return out, sse, sigma_est

    def denoise(self, noisy, sigma_est=0):
        z = _as_dev(noisy, self.dtype)
        H, W = z.shape[-2:]
        s = torch.full((1,), float(sigma_est), dtype=z.dtype, device=z.device)
        out, _, _ = self.denoise_device(z.reshape(1, H, W), sigma_est=s)
        if isinstance(noisy, torch.Tensor):
            return out.reshape(noisy.shape)
        return out.reshape(H, W).double().cpu().numpy()


def mmo_weights_from_state_dict(sd, negative_slope=0.01):
    """`simple_CNN` state dict (reference MMODenoise.py:80-82: in_conv, conv_list.{i}, out_conv, each with a bias;
    an optional DataParallel 'module.' prefix is dropped) -> the weight dict `ops.DncnnPlan` takes.  The reference
    feeds the network the transposed image (MMODenoise.py:124), which is the same as transposing every 3x3 kernel."""
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
    n_mid = len([k for k in sd if k.startswith('conv_list.') and k.endswith('.weight')])
    names = ['in_conv'] + [f'conv_list.{i}' for i in range(n_mid)] + ['out_conv']
    out = {'n_layers': np.int64(len(names)), 'negative_slope': float(negative_slope), 'transpose_taps': True}
    for i, n in enumerate(names):
        out[f'conv{i}.weight'] = np.asarray(sd[n + '.weight'].detach().cpu().numpy() if hasattr(sd[n + '.weight'], 'detach') else sd[n + '.weight'])
        out[f'conv{i}.bias'] = np.asarray(sd[n + '.bias'].detach().cpu().numpy() if hasattr(sd[n + '.bias'], 'detach') else sd[n + '.bias'])
    return out


class MMODenoiser(Denoise):
    """reference denoisers/MMODenoise.py:105-128: the 20-layer bias/LeakyReLU/skip network `simple_CNN` (:73-101) on
    the MFMA conv stack.  `model` may be the reference's torch module (plain or DataParallel) or a
    state dict; `weights=` takes a ready weight dict.  With neither, the reference's checkpoint path is tried with
    `weights_only=True`: the files the reference ships are whole pickled modules, which that loader refuses by design
    -- re-save `model.module.state_dict()` once and pass the file as `path=`.  Single-channel images only.
    Like the reference: `t` advances per call, `sigma_est` is ignored, the result is float32 in [0, 1]."""

    def __init__(self, model=None, channels=3, path=None, cuda=True, sigma=0.01, root_path='.', *, weights=None):
        super().__init__()
        self.sigma = sigma
        if channels != 1:
            raise NotImplementedError('MMODenoiser on the MI355X path handles single-channel (2-D) images: channels=1')
        if weights is None:
            if model is None:
                pth = path if path is not None else (root_path + 'checkpoints/pretrained/DnCNN_nobn_nch_' + str(channels)
                                                     + '_nlev_' + str(sigma) + '.pth')
                model = torch.load(pth, map_location='cpu', weights_only=True)   # FileNotFoundError like the reference
            if hasattr(model, 'module'):
                model = model.module
            sd = model.state_dict() if hasattr(model, 'state_dict') else model
            weights = mmo_weights_from_state_dict(sd)
        else:
            weights = dict(weights)
            weights.setdefault('transpose_taps', True)
            weights.setdefault('negative_slope', 0.01)
        self.model = weights
        self._plans = {}

    def _plan(self, B, H, W):
        key = (B, H, W)
        if key not in self._plans:
            self._plans[key] = ops.DncnnPlan(self.model, H, W, B)
        return self._plans[key]

    def denoise_device(self, z, sigma_est=None, xrec=None, out=None, sse=None):
        self.t += 1
        B, H, W = z.shape
        out, sse = self._plan(B, H, W).mmo_denoise(z, xrec=xrec, out=out, sse=sse)
        return out, sse, None

    def denoise(self, noisy, sigma_est=0):
        z = _as_dev(noisy)
        H, W = z.shape[-2:]
        out, _, _ = self.denoise_device(z.reshape(1, H, W))
        if isinstance(noisy, torch.Tensor):
            return out.reshape(noisy.shape)
        return out.reshape(H, W).float().cpu().numpy()
