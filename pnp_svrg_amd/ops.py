"""Torch-tensor front end of the C-ABI calls.  PyTorch is plumbing here: it owns HBM
allocations and the HIP stream; every op below is one call into libpnp_hip.so."""
import ctypes
import torch
from . import _native as N

_DT = {torch.float32: N.F32, torch.float64: N.F64}
_CDT = {torch.float32: torch.complex64, torch.float64: torch.complex128}


def _stream():
    """The current HIP stream of the current device as a raw pointer (capture-aware: inside torch.cuda.graph it is the
    capture stream).  Through torch._C directly: torch.cuda.current_stream() costs 5-6 us of Python per call, which at five
    launches per inner iteration is as much as a kernel of the B = 1 loops."""
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:                                      # (private API moved: the public, slower way)
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'device-resident contiguous tensors only'
    return ctypes.c_void_p(t.data_ptr())


def require_gpu():
    if not torch.cuda.is_available():
        raise N.NativeError('no MI355X visible (torch.cuda.is_available() is False); the hot path has no CPU fallback')
    N.lib()


class CsmriPlan:
    """pnp_csmri_plan_* : masked-FFT gradient of B independent H x W CSMRI problems."""

    def __init__(self, H, W, batch, dtype=torch.float32):
        require_gpu()
        self.H, self.W, self.B, self.dtype = H, W, batch, dtype
        h = ctypes.c_void_p()
        N.call('pnp_csmri_plan_create', ctypes.byref(h), H, W, batch, _DT[dtype])
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                N.lib().pnp_csmri_plan_destroy(h)
            except Exception:
                pass

    def sel_from_indices(self, idx, out=None):
        """idx: int32 [B, n] flat row-major k-space positions -> uint8 [B, W, H] (transposed)."""
        assert idx.dtype == torch.int32 and idx.shape[0] == self.B
        out = out if out is not None else torch.empty((self.B, self.W, self.H), dtype=torch.uint8, device=idx.device)
        N.call('pnp_csmri_sel_from_indices', self._h, _p(idx), idx.shape[1], _p(out), _stream())
        return out

    def pack_mask(self, selT, out=None):
        """uint8 transposed selector [B, W, H] -> bit-packed mask int32 [B, W, H/32] (pnp_csmri_pack_mask)."""
        assert selT.dtype == torch.uint8 and tuple(selT.shape) == (self.B, self.W, self.H)
        out = out if out is not None else torch.empty((self.B, self.W, self.H // 32), dtype=torch.int32, device=selT.device)
        N.call('pnp_csmri_pack_mask', self._h, _p(selT), _p(out), _stream())
        return out

    def draw_thresholds(self, bits, mb, seed, step0, nsteps=1, out=None, step_dev=None, selbits=None):
        """Device-side minibatch draws for steps step0 .. step0+nsteps-1 of every problem: threshold descriptors
        (int64 [nsteps, B, 2] = 16 bytes per (step, problem)) and, when `selbits` (int32 [nsteps, B, W, H/32]) is given,
        mask o minibatch as bit-packed selectors -- `grad(bits=selbits[j])` consumes a step's row."""
        assert bits.dtype == torch.int32 and tuple(bits.shape) == (self.B, self.W, self.H // 32)
        out = out if out is not None else torch.empty((nsteps, self.B, 2), dtype=torch.int64, device=bits.device)
        assert out.dtype == torch.int64 and tuple(out.shape) == (nsteps, self.B, 2)
        assert selbits is None or (selbits.dtype == torch.int32 and tuple(selbits.shape) == (nsteps, self.B, self.W, self.H // 32))
        N.call('pnp_csmri_draw_thresholds', self._h, _p(bits), int(mb), int(seed) & (2 ** 64 - 1), int(step0) & 0xFFFFFFFF,
               int(nsteps), _p(step_dev), _p(out), _p(selbits), _stream())
        return out

    def sel_from_thresholds(self, bits, mbd, out=None):
        """mask o minibatch of one step (mbd: int64 [B, 2]) as an explicit transposed uint8 selector."""
        assert mbd.dtype == torch.int64 and tuple(mbd.shape) == (self.B, 2)
        out = out if out is not None else torch.empty((self.B, self.W, self.H), dtype=torch.uint8, device=bits.device)
        N.call('pnp_csmri_sel_from_thresholds', self._h, _p(bits), _p(mbd), _p(out), _stream())
        return out

    def draw_minibatch(self, bits, mb, seed, step, out=None, step_dev=None):
        """Device-side uniform draw of `mb` of each problem's sampled locations -> explicit transposed selector.
        bits: int32 [B, W, H/32] bit-packed mask (pack_mask)."""
        assert bits.dtype == torch.int32 and tuple(bits.shape) == (self.B, self.W, self.H // 32)
        out = out if out is not None else torch.empty((self.B, self.W, self.H), dtype=torch.uint8, device=bits.device)
        N.call('pnp_csmri_draw_minibatch', self._h, _p(bits), int(mb), int(seed) & (2 ** 64 - 1), int(step) & 0xFFFFFFFF,
               _p(step_dev), _p(out), _stream())
        return out

    def sel_from_dense(self, sel, out=None):
        assert sel.dtype == torch.uint8 and tuple(sel.shape) == (self.B, self.H, self.W)
        out = out if out is not None else torch.empty((self.B, self.W, self.H), dtype=torch.uint8, device=sel.device)
        N.call('pnp_csmri_sel_from_dense', self._h, _p(sel), _p(out), _stream())
        return out

    def pack_y(self, YT, selT, out=None):
        """YT: complex [B, W, H] (Y transposed); returns packed data term complex [B, W/2, H]."""
        assert YT.dtype == _CDT[self.dtype] and tuple(YT.shape) == (self.B, self.W, self.H)
        out = out if out is not None else torch.empty((self.B, self.W // 2, self.H), dtype=YT.dtype, device=YT.device)
        N.call('pnp_csmri_pack_y', self._h, _p(YT), _p(selT), _p(out), _stream())
        return out

    def grad(self, a, selT=None, b=None, yh=None, alpha=1.0, beta=0.0, c1=None, gamma=0.0, c2=None, out=None, *,
             bits=None, alpha_vec=None, YT=None):
        """out = alpha * alpha_vec[b] * Re ifft2(sel o fft2(a - b) - sel o Y) + beta*c1 + gamma*c2.
        Selector: `selT` (explicit uint8 [B, W, H]) or `bits` (bit-packed int32 [B, W, H/32]: the mask, or one step's
        row of draw_thresholds' selbits).  Data term: `yh` (packed for this
        selector) or `YT` (complex [B, W, H], masked by the selector inside the kernel)."""
        for t in (a, b, c1, c2, out):
            assert t is None or (t.dtype == self.dtype and t.numel() == self.B * self.H * self.W)
        assert (selT is None) != (bits is None), 'pass selT or bits'
        assert alpha_vec is None or (alpha_vec.dtype == self.dtype and alpha_vec.numel() == self.B)
        out = out if out is not None else torch.empty_like(a)
        assert YT is None or (YT.dtype == _CDT[self.dtype] and tuple(YT.shape) == (self.B, self.W, self.H))
        N.call('pnp_csmri_grad_sel', self._h, _p(a), _p(b), _p(selT), _p(bits), _p(yh), _p(YT), float(alpha), _p(alpha_vec),
               float(beta), _p(c1), float(gamma), _p(c2), _p(out), _stream())
        return out


    def svrg_step(self, a, b, bits, alpha=1.0, beta=0.0, c1=None, gamma=0.0, c2=None, out=None, *, alpha_vec=None, denoise=True,
                  sigma_modifier=1.0, fallback_sigma=0.0, xrec=None, sse=None, sigma_out=None):
        """pnp_csmri_svrg_step: gradient step + noise estimate (+ TV prox) + PSNR error of one inner iteration in ONE
        kernel (f32, 256 x 256).  out = prox_TV(alpha * Re ifft2(bits o fft2(a - b)) + beta*c1 + gamma*c2)."""
        assert self.dtype == torch.float32 and self.H == 256 and self.W == 256
        assert bits.dtype == torch.int32 and tuple(bits.shape) == (self.B, self.W, self.H // 32)
        out = out if out is not None else torch.empty_like(a)
        sigma_out = sigma_out if sigma_out is not None else torch.empty(self.B, dtype=a.dtype, device=a.device)
        N.call('pnp_csmri_svrg_step', self._h, _p(a), _p(b), _p(bits), float(alpha), _p(alpha_vec), float(beta), _p(c1),
               float(gamma), _p(c2), _p(out), 1 if denoise else 0, float(sigma_modifier), float(fallback_sigma), _p(xrec),
               _p(sse), _p(sigma_out), _stream())
        return out, sse, sigma_out


    def svrg_outer_step(self, z, mask_bits, yh, alpha_vec, lr, w_out, mu_out, out=None, *, denoise=True, sigma_modifier=1.0,
                        fallback_sigma=0.0, xrec=None, sse=None, sigma_out=None):
        """pnp_csmri_svrg_outer_step: the outer refresh of pnp_svrg folded into its first inner iteration, one kernel:
        mu_out = alpha_vec[b] * Re ifft2(mask o fft2(z) - yh), w_out = z, out = prox_TV(z - lr * mu_out)."""
        assert self.dtype == torch.float32 and self.H == 256 and self.W == 256
        assert mask_bits.dtype == torch.int32 and tuple(mask_bits.shape) == (self.B, self.W, self.H // 32)
        for t in (z, w_out, mu_out, out):
            assert t is None or (t.dtype == self.dtype and t.numel() == self.B * self.H * self.W)
        out = out if out is not None else torch.empty_like(z)
        sigma_out = sigma_out if sigma_out is not None else torch.empty(self.B, dtype=z.dtype, device=z.device)
        N.call('pnp_csmri_svrg_outer_step', self._h, _p(z), _p(mask_bits), _p(yh), _p(alpha_vec), float(lr), _p(w_out), _p(mu_out),
               _p(out), 1 if denoise else 0, float(sigma_modifier), float(fallback_sigma), _p(xrec), _p(sse), _p(sigma_out), _stream())
        return out, sse, sigma_out


    def svrg_outer_iteration(self, z, w, mu, mask_bits, yh, alpha_vec, selbits, T2, lr, mini_batch_size, xrec, sse_log, log_row0,
                             sigma_out, *, sigma_modifier=1.0, fallback_sigma=0.0):
        """pnp_csmri_svrg_outer_iteration: refresh + T2 inner iterations with the TV prox in ONE launch (z in place; w, mu out)."""
        assert self.dtype == torch.float32 and self.H == 256 and self.W == 256
        for t in (z, w, mu, xrec):
            assert t.dtype == self.dtype and t.numel() == self.B * self.H * self.W
        assert selbits.dtype == torch.int32 and tuple(selbits.shape) == (T2, self.B, self.W, self.H // 32)
        assert sse_log.dtype == torch.float64 and sse_log.dim() == 2 and sse_log.shape[1] == self.B and sse_log.is_contiguous()
        N.call('pnp_csmri_svrg_outer_iteration', self._h, _p(z), _p(w), _p(mu), _p(mask_bits), _p(yh), _p(alpha_vec), _p(selbits), int(T2),
               float(lr), int(mini_batch_size), float(sigma_modifier), float(fallback_sigma), _p(xrec), _p(sse_log), int(log_row0),
               int(sse_log.shape[0]), _p(sigma_out), _stream())


class DncnnPlan:
    """pnp_dncnn_plan_*: DnCNN-17 prox for B images of H x W (fp32 network on the f32 matrix cores).

    `weights`: dict of NumPy arrays in the reference's layer order -- conv{i}.weight (i = 0..n-1) and
    bn{i}.{weight,bias,mean,var} for the middle layers -- exactly what tests/golden/dncnn_noise15.npz
    holds or what `load_dncnn_state_dict` extracts from a reference .pth.  BatchNorm (eval) is folded here.
    Optional `conv{i}.bias` arrays (any layer), `negative_slope` (LeakyReLU instead of ReLU) and
    `transpose_taps` (every 3x3 kernel transposed) cover the MMO `simple_CNN` (denoisers/MMODenoise.py:73-101)."""

    def __init__(self, weights, H, W, batch, winograd=None):
        """winograd: 5 = Winograd F(4x4,3x3) conv kernel (default where H % 8 == 0 and W % 64 == 0; fp32, a quarter of the
        matrix-core work), True / 1 = F(2,3) along x (two thirds; the default elsewhere), False / 0 = direct implicit GEMM
        (bit-for-bit an fmaf chain), 6 = opt-in F(4x4,3x3) on the bf16 matrix cores with three-way exact splits (fp32-class accuracy,
        currently slower than 5), None = env PNP_DNCNN_WINOGRAD or the default."""
        import numpy as np
        require_gpu()
        n = int(weights['n_layers'])
        self.H, self.W, self.B, self.n_mid = H, W, batch, n - 2
        tr = bool(weights.get('transpose_taps', False))

        def taps(name, shape):
            w = np.asarray(weights[name], dtype=np.float64).reshape(shape + (3, 3))
            return (w.swapaxes(-1, -2) if tr else w).reshape(shape + (9,))

        w_first = np.ascontiguousarray(taps('conv0.weight', (64,)), dtype=np.float32)
        w_last = np.ascontiguousarray(taps(f'conv{n - 1}.weight', (64,)), dtype=np.float32)
        w_mid = np.empty((n - 2, 64, 64, 9), dtype=np.float32)
        b_mid = np.zeros((n - 2, 64), dtype=np.float32)
        for i in range(1, n - 1):
            w = taps(f'conv{i}.weight', (64, 64))
            b = np.asarray(weights[f'conv{i}.bias'], np.float64) if f'conv{i}.bias' in weights else np.zeros(64)
            if f'bn{i}.weight' in weights:
                s = np.asarray(weights[f'bn{i}.weight'], np.float64) / np.sqrt(np.asarray(weights[f'bn{i}.var'], np.float64) + 1e-5)
                w = w * s[:, None, None]
                b = np.asarray(weights[f'bn{i}.bias'], np.float64) + (b - np.asarray(weights[f'bn{i}.mean'], np.float64)) * s
            b_mid[i - 1] = b
            w_mid[i - 1] = w
        h = ctypes.c_void_p()
        N.call('pnp_dncnn_plan_create', ctypes.byref(h), n - 2, w_first.ctypes.data_as(ctypes.c_void_p),
               w_mid.ctypes.data_as(ctypes.c_void_p), b_mid.ctypes.data_as(ctypes.c_void_p),
               w_last.ctypes.data_as(ctypes.c_void_p), H, W, batch)
        self._h = h
        slope = float(weights.get('negative_slope', 0.0))
        if slope != 0.0 or 'conv0.bias' in weights or f'conv{n - 1}.bias' in weights:
            b_first = np.ascontiguousarray(weights.get('conv0.bias', np.zeros(64)), dtype=np.float32)
            b_last = float(np.asarray(weights.get(f'conv{n - 1}.bias', 0.0)).reshape(-1)[0])
            N.call('pnp_dncnn_set_affine', self._h, b_first.ctypes.data_as(ctypes.c_void_p), b_last, slope)
        if winograd is not None:
            N.call('pnp_dncnn_set_winograd', self._h, int(winograd))

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                N.lib().pnp_dncnn_plan_destroy(h)
            except Exception:
                pass

    def profile_begin(self, max_calls=4096):
        N.call('pnp_dncnn_profile_begin', self._h, int(max_calls))

    def profile_end(self):
        """-> (mean ms of one 64->64 conv launch, number of launches timed)"""
        ms, n = ctypes.c_double(), ctypes.c_long()
        N.call('pnp_dncnn_profile_end', self._h, ctypes.byref(ms), ctypes.byref(n))
        return ms.value, n.value

    def debug_w44_weights(self, layer):
        """packed F(4x4,3x3) weights of one middle layer as a device tensor (test hook)"""
        n = N.lib().pnp_dncnn_debug_w44_floats()
        dst = torch.empty(n, dtype=torch.float32, device='cuda')
        N.call('pnp_dncnn_debug_w44_weights', self._h, int(layer), _p(dst), _stream())
        return dst

    def debug_mid_layer(self, layer, x, out, w44=None, rows=0):
        """one 64->64 layer on caller-provided activations [B,64,H,W] (test hook: guard bands around the buffers)"""
        assert x.dtype == torch.float32 and tuple(x.shape) == (self.B, 64, self.H, self.W) and tuple(out.shape) == tuple(x.shape)
        N.call('pnp_dncnn_debug_mid_layer', self._h, int(layer), _p(x), _p(out), _p(w44), int(rows), _stream())
        return out

    def forward(self, x, out=None):
        """raw network residual; x: float32 [B,H,W]."""
        assert x.dtype == torch.float32 and tuple(x.shape) == (self.B, self.H, self.W)
        out = out if out is not None else torch.empty_like(x)
        N.call('pnp_dncnn_forward', self._h, _p(x), _p(out), _stream())
        return out

    def denoise(self, z, sigma_net, xrec=None, out=None, sse=None):
        assert tuple(z.shape) == (self.B, self.H, self.W)
        out = out if out is not None else torch.empty_like(z)
        if xrec is not None and sse is None:
            sse = torch.empty(self.B, dtype=torch.float64, device=z.device)
        N.call('pnp_dncnn_denoise', self._h, _p(z), _p(out), _DT[z.dtype], float(sigma_net), _p(xrec), _p(sse), _stream())
        return out, sse


    def mmo_denoise(self, z, xrec=None, out=None, sse=None):
        """pnp_mmo_denoise: clip(xc + net(xc), 0, 1), xc = clip(z, 0, 1) (reference MMODenoise.py:18-40,88-101)."""
        assert tuple(z.shape) == (self.B, self.H, self.W)
        out = out if out is not None else torch.empty_like(z)
        if xrec is not None and sse is None:
            sse = torch.empty(self.B, dtype=torch.float64, device=z.device)
        N.call('pnp_mmo_denoise', self._h, _p(z), _p(out), _DT[z.dtype], _p(xrec), _p(sse), _stream())
        return out, sse


def sigma_est(z):
    """z: [B, H, W] -> [B] (estimate_sigma(multichannel=True, average_sigmas=True))."""
    require_gpu()
    B, H, W = z.shape
    out = torch.empty(B, dtype=z.dtype, device=z.device)
    N.call('pnp_sigma_est', _p(z), H, W, B, _DT[z.dtype], _p(out), _stream())
    return out


def prox_tv(z, sigma_in=None, sigma_modifier=1.0, fallback_sigma=0.0, xrec=None, out=None, sse=None, sigma_out=None):
    """Fused estimate_sigma + Haar-BayesShrink prox (+ squared error vs xrec).
    Returns (denoised [B,H,W], sse [B] float64 or None, sigma_est [B])."""
    require_gpu()
    B, H, W = z.shape
    out = out if out is not None else torch.empty_like(z)
    if xrec is not None and sse is None:
        sse = torch.empty(B, dtype=torch.float64, device=z.device)
    sigma_out = sigma_out if sigma_out is not None else torch.empty(B, dtype=z.dtype, device=z.device)
    N.call('pnp_prox_tv', _p(z), _p(out), H, W, B, _DT[z.dtype], _p(sigma_in), float(sigma_modifier),
           float(fallback_sigma), _p(xrec), _p(sse), _p(sigma_out), _stream())
    return out, sse, sigma_out


def sse(z, xrec, out=None):
    require_gpu()
    B = z.shape[0]
    out = out if out is not None else torch.empty(B, dtype=torch.float64, device=z.device)
    N.call('pnp_sse', _p(z), _p(xrec), z.numel() // B, B, _DT[z.dtype], _p(out), _stream())
    return out


def minmax(z):
    require_gpu()
    B = z.shape[0]
    out = torch.empty((B, 2), dtype=z.dtype, device=z.device)
    N.call('pnp_minmax', _p(z), z.numel() // B, B, _DT[z.dtype], _p(out), _stream())
    return out


def axpbypcz(a, x, b=0.0, y=None, c=0.0, w=None, out=None):
    require_gpu()
    out = out if out is not None else torch.empty_like(x)
    N.call('pnp_axpbypcz', float(a), _p(x), float(b), _p(y), float(c), _p(w), _p(out), x.numel(), _DT[x.dtype], _stream())
    return out


_NLM_W0 = {}


def _nlm_w0(side, device):
    """exp(-(x^2+y^2)/(2A^2)), A=(side-1)/4, and its NumPy sum (skimage non_local_means.py:150-153)."""
    key = (side, str(device))
    if key not in _NLM_W0:
        import numpy as np
        off = side // 2
        A = (side - 1.0) / 4.0
        g = np.arange(-off, off + 1)
        gr, gc = np.meshgrid(g, g, indexing='ij')
        w = np.exp(-(gr * gr + gc * gc) / (2 * A * A))
        _NLM_W0[key] = (torch.from_numpy(np.ascontiguousarray(w.ravel())).to(device), float(np.sum(w)))
    return _NLM_W0[key]


def nlm2d(z, sigma_in=None, sigma_modifier=1.0, fixed_h=0.0, patch_size=4, patch_distance=5, xrec=None, out=None, sse=None):
    """skimage denoise_nl_means(slow mode) semantics on [B,H,W]; returns (denoised, sse or None)."""
    require_gpu()
    B, H, W = z.shape
    side = patch_size + 1 if patch_size % 2 == 0 else patch_size
    w0, w0_sum = _nlm_w0(side, z.device)
    out = out if out is not None else torch.empty_like(z)
    ws = None
    if xrec is not None:
        sse = sse if sse is not None else torch.empty(B, dtype=torch.float64, device=z.device)
        ws = torch.empty(B * ((H + 15) // 16) * ((W + 15) // 16), dtype=torch.float64, device=z.device)
    N.call('pnp_nlm2d', _p(z), _p(out), H, W, B, _DT[z.dtype], int(patch_size), int(patch_distance), _p(sigma_in),
           float(sigma_modifier), float(fixed_h), _p(w0), w0_sum, _p(xrec), _p(sse), _p(ws), _stream())
    return out, sse


class DeblurPlan:
    """pnp_deblur_plan_*: B^T S^T (S B z - y) for B problems of H x W (H*W in {4096, 65536})."""

    def __init__(self, H, W, batch, dtype, Bk, bilinear=None):
        import numpy as np
        require_gpu()
        self.H, self.W, self.N, self.B, self.dtype = H, W, H * W, batch, dtype
        npdt = np.float32 if dtype == torch.float32 else np.float64
        Bk = np.ascontiguousarray(Bk, dtype=npdt)
        h = ctypes.c_void_p()
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        if bilinear is None:
            self.M = self.N
            N.call('pnp_deblur_plan_create', ctypes.byref(h), H, W, batch, _DT[dtype], vp(Bk), self.M, None, None, None, None, None)
        else:
            g_idx, g_w, a_rowptr, a_col, a_val = bilinear
            self.M = g_idx.shape[0]
            keep = [np.ascontiguousarray(g_idx, np.int32), np.ascontiguousarray(g_w, npdt), np.ascontiguousarray(a_rowptr, np.int32),
                    np.ascontiguousarray(a_col, np.int32), np.ascontiguousarray(a_val, npdt)]
            N.call('pnp_deblur_plan_create', ctypes.byref(h), H, W, batch, _DT[dtype], vp(Bk), self.M, *[vp(k) for k in keep])
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                N.lib().pnp_deblur_plan_destroy(h)
            except Exception:
                pass

    def grad(self, z, Y, sel=None, scale=1.0, out=None, mbd=None):
        """scale * B^T S^T (sel o (S B z - Y)); sel: uint8 [B, M] indicator, or mbd: int64 [B, 2] = one step's
        device-drawn minibatch descriptors (draw_thresholds), or neither (all measurements)."""
        assert z.dtype == self.dtype and z.numel() == self.B * self.N and Y.numel() == self.B * self.M
        assert sel is None or mbd is None
        out = out if out is not None else torch.empty_like(z)
        if mbd is not None:
            assert mbd.dtype == torch.int64 and tuple(mbd.shape) == (self.B, 2)
            N.call('pnp_deblur_grad_mb', self._h, _p(z), _p(Y), _p(mbd), float(scale), _p(out), _stream())
        else:
            N.call('pnp_deblur_grad', self._h, _p(z), _p(Y), _p(sel), float(scale), _p(out), _stream())
        return out

    def forward(self, x, out=None):
        out = out if out is not None else torch.empty(self.B * self.M, dtype=self.dtype, device=x.device)
        N.call('pnp_deblur_forward', self._h, _p(x), _p(out), _stream())
        return out


def pr_grad(A, w, y, rows=None, scale=1.0, workspace=None, out=None):
    """scale * A_sel^T(((|A_sel w| - y_sel)/|A_sel w|) o A_sel w); A [M,N], rows int32 [nsel] or None."""
    require_gpu()
    M, Nn = A.shape
    if workspace is None:
        workspace = torch.empty(N.lib().pnp_pr_workspace_elems(M, Nn), dtype=A.dtype, device=A.device)
    out = out if out is not None else torch.empty(Nn, dtype=A.dtype, device=A.device)
    nsel = M if rows is None else rows.numel()
    N.call('pnp_pr_grad', _p(A), _p(w), _p(y), _p(rows), nsel, M, Nn, _DT[A.dtype], float(scale), _p(workspace), _p(out), _stream())
    return out


def pr_grad_batch(A, w, y, rows=None, scale=1.0, workspace=None, out=None):
    """B independent problems: A [B,M,N], w [B,N], y [B,M], rows int32 [B,nsel] or None -> [B,N]."""
    require_gpu()
    B, M, Nn = A.shape
    if workspace is None:
        workspace = torch.empty(B * N.lib().pnp_pr_workspace_elems(M, Nn), dtype=A.dtype, device=A.device)
    out = out if out is not None else torch.empty((B, Nn), dtype=A.dtype, device=A.device)
    nsel = M if rows is None else rows.shape[1]
    N.call('pnp_pr_grad_batch', _p(A), _p(w), _p(y), _p(rows), nsel, M, Nn, B, _DT[A.dtype], float(scale), _p(workspace),
           _p(out), _stream())
    return out


def draw_thresholds(M, B, mb, seed, step0, nsteps=1, out=None, step_dev=None, device='cuda'):
    """Device-side draws of `mb` of M measurements for B problems and `nsteps` steps -> descriptors int64 [nsteps, B, 2]."""
    require_gpu()
    out = out if out is not None else torch.empty((nsteps, B, 2), dtype=torch.int64, device=device)
    N.call('pnp_draw_thresholds', int(M), int(B), int(mb), int(seed) & (2 ** 64 - 1), int(step0) & 0xFFFFFFFF, int(nsteps),
           _p(step_dev), _p(out), _stream())
    return out


def indicator_from_thresholds(M, mbd, out=None):
    require_gpu()
    B = mbd.shape[0]
    out = out if out is not None else torch.empty((B, M), dtype=torch.uint8, device=mbd.device)
    N.call('pnp_indicator_from_thresholds', int(M), int(B), _p(mbd), _p(out), _stream())
    return out


def rows_from_thresholds(M, mb, mbd, out=None):
    require_gpu()
    B = mbd.shape[0]
    out = out if out is not None else torch.empty((B, mb), dtype=torch.int32, device=mbd.device)
    N.call('pnp_rows_from_thresholds', int(M), int(B), int(mb), _p(mbd), _p(out), _stream())
    return out


def indicator_from_indices(idx, M, out=None):
    """idx int32 [B, n] -> uint8 [B, M] (Problem.select_mb's 0/1 indicator, problems/problem.py:110-117)."""
    require_gpu()
    assert idx.dtype == torch.int32
    B, n = idx.shape
    out = out if out is not None else torch.empty((B, M), dtype=torch.uint8, device=idx.device)
    N.call('pnp_indicator_from_indices', _p(idx), int(n), int(M), int(B), _p(out), _stream())
    return out


def saga_table_update(z, g, slot, prev, tsum, lr, inv_hist):
    """pnp_saga_table_update: one SAGA step over all elements (z, slot, tsum updated in place)."""
    require_gpu()
    for t in (g, slot, prev, tsum):
        assert t.dtype == z.dtype and t.numel() == z.numel()
    N.call('pnp_saga_table_update', _p(z), _p(g), _p(slot), _p(prev), _p(tsum), float(lr), float(inv_hist), z.numel(),
           _DT[z.dtype], _stream())


def pr_spectral_apply(A, v, y, scale=1.0, workspace=None, out=None):
    """scale * A^T (y o (A v)): one power-iteration step of the PR spectral initialisation; A [M,N]."""
    require_gpu()
    M, Nn = A.shape
    if workspace is None:
        workspace = torch.empty(N.lib().pnp_pr_workspace_elems(M, Nn), dtype=A.dtype, device=A.device)
    out = out if out is not None else torch.empty(Nn, dtype=A.dtype, device=A.device)
    N.call('pnp_pr_spectral_apply', _p(A), _p(v), _p(y), M, Nn, _DT[A.dtype], float(scale), _p(workspace), _p(out), _stream())
    return out


def counter_add(counter, inc=1):
    """counter: int32 device tensor with one element (device-resident step counter)."""
    require_gpu()
    N.call('pnp_counter_add', _p(counter), int(inc), _stream())


def log_append(src, log, step_dev):
    """log[(step % n_log)] = src; src: float64 [n], log: float64 [n_log, n], step_dev: device counter."""
    require_gpu()
    N.call('pnp_log_append', _p(src), src.numel(), _p(log), log.shape[0], _p(step_dev), _stream())


def log_append_inc(src, log, counter):
    """log[(counter % n_log)] = src; counter += 1 (one launch; counter: int32 device tensor with one element)."""
    require_gpu()
    N.call('pnp_log_append_inc', _p(src), src.numel(), _p(log), log.shape[0], _p(counter), _stream())
