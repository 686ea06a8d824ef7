"""Batched, sync-free PnP-SVRG engine: B independent CSMRI reconstructions advance together,
one inner iteration per `step()`, entirely in HBM.

This is the throughput form of reference algorithms/pnp_svrg.py:26-95 (outer: mu = grad_full(z),
w = z; inner: minibatch draw, SVRG direction, step, estimate_sigma, denoise, PSNR log) for the
sweep-style workloads of script_diff_*_set12.py, where many reconstructions are independent.
Nothing is read back per iteration: squared errors (for PSNR) accumulate in a device log.

Minibatches: by default drawn on the device inside each step (`pnp_csmri_draw_minibatch`: hash keys +
radix select, uniform without replacement); for reference-identical runs pass index lists drawn from
the legacy `np.random` stream (`step(idx)`).
"""
import numpy as np
import torch

from . import ops


class CsmriBatch:
    """Device-resident data of B CSMRI problems (reference problems/CSMRI.py:12-41 per problem)."""

    def __init__(self, xrec, mask, Y, xinit, dtype=torch.float32, device='cuda'):
        ops.require_gpu()
        B, H, W = xrec.shape
        self.B, self.H, self.W, self.N, self.dtype = B, H, W, H * W, dtype
        cdt = torch.complex64 if dtype == torch.float32 else torch.complex128
        self.plan = ops.CsmriPlan(H, W, B, dtype)
        self.xrec = torch.from_numpy(np.ascontiguousarray(xrec, np.float64)).to(device, dtype)
        self.xinit = torch.from_numpy(np.ascontiguousarray(xinit, np.float64)).to(device, dtype).reshape(B, H, W)
        self.M0 = mask.reshape(B, -1).sum(1).astype(np.int64)
        mask_u8 = torch.from_numpy(np.ascontiguousarray(mask, np.uint8)).to(device)
        self.maskT = self.plan.sel_from_dense(mask_u8)
        YT = torch.from_numpy(np.ascontiguousarray(np.swapaxes(Y, 1, 2))).to(device, cdt).contiguous()
        self.yh_full = self.plan.pack_y(YT, self.maskT)
        self._Y_np = Y                                          # host copy: the fused-TV engine packs the transposed problem's data term
        self.inv_m0 = None
        if not np.all(self.M0 == self.M0[0]):
            # per-problem 1/M0 differs: fold it into the data (grad_full is linear in 1/M0)
            raise ValueError('CsmriBatch needs the same number of sampled k-space points in every problem; '
                             'use synthetic(), which draws masks with a fixed count, or batch equal-M0 problems')
        self.mask_np = np.ascontiguousarray(mask, np.uint8)
        # flatnonzero(mask) per problem (equal counts) for the device-side minibatch draw
        self.mask_idx = torch.from_numpy(np.stack([np.flatnonzero(m) for m in self.mask_np.reshape(B, -1)]).astype(np.int32)).to(device)

    def Y_dev(self, cdt):
        """the k-space data, un-transposed, on the device (complex [B, H, W])"""
        return torch.from_numpy(np.ascontiguousarray(self._Y_np)).to(self.xrec.device, cdt).contiguous()

    @classmethod
    def synthetic(cls, B, H=256, W=256, sample_prob=0.2, snr=20.0, seed=0, dtype=torch.float32):
        """B synthetic problems: smoothed-noise images, masks with exactly
        round(p*N) sampled points (so 1/M0 is shared), complex data with real noise on the support."""
        rng = np.random.default_rng(seed)
        N = H * W
        m0 = int(round(sample_prob * N))
        xrec = np.empty((B, H, W))
        mask = np.zeros((B, N), np.uint8)
        Y = np.empty((B, H, W), np.complex128)
        xinit = np.empty((B, N))
        for b in range(B):
            x = rng.random((H, W))
            p = np.pad(x, 2, mode='wrap')
            y = sum(p[i:i + H, j:j + W] for i in range(5) for j in range(5)) / 25.0
            y = (y - y.min()) / (y.max() - y.min())
            xrec[b] = np.round(y * 255) / 255.0
            xrec[b] = (xrec[b] - xrec[b].min()) / (xrec[b].max() - xrec[b].min())
            mask[b, rng.choice(N, m0, replace=False)] = 1
            mk = mask[b].reshape(H, W)
            Y0 = mk * np.fft.fft2(xrec[b])
            sigma = np.sqrt(np.linalg.norm(Y0.ravel()) / 10 ** (snr / 10) / H / W)     # problem.py:58-61
            Y[b] = Y0 + mk * rng.normal(0, sigma, (H, W))
            xi = np.absolute(np.fft.ifft2(Y[b])).ravel()
            xinit[b] = (xi - xi.min()) / (xi.max() - xi.min())
        return cls(xrec, mask.reshape(B, H, W), Y, xinit, dtype=dtype)

    def draw_minibatches(self, n_steps, mb, seed=1):
        """[n_steps][B][mb] int32 flat k-space indices, each row a uniform draw without replacement
        from that problem's mask support (CSMRI.py:66-74 semantics, fast Generator stream)."""
        rng = np.random.default_rng(seed)
        out = np.empty((n_steps, self.B, mb), np.int32)
        for b in range(self.B):
            locs = np.flatnonzero(self.mask_np[b]).astype(np.int32)
            for s in range(n_steps):
                out[s, b] = rng.choice(locs, mb, replace=False)
        return torch.from_numpy(out).to(self.xrec.device)


class TVProx:
    """denoisers/TV.py semantics for the engine (fused estimate_sigma + BayesShrink + error sum)."""

    def __init__(self, sigma_modifier=1.0, decay=1.0, denoise_strength=0.0):
        self.sigma_modifier, self.decay, self.denoise_strength, self.t = sigma_modifier, decay, denoise_strength, 0

    def bind(self, batch):
        self.sig = torch.empty(batch.B, dtype=batch.dtype, device=batch.xrec.device)

    def __call__(self, z, xrec, sse_out):
        self.t += 1
        ops.prox_tv(z, sigma_modifier=self.sigma_modifier, fallback_sigma=self.denoise_strength * self.decay ** self.t,
                    xrec=xrec, out=z, sse=sse_out, sigma_out=self.sig)


class DnCNNProx:
    """denoisers/RealSN_DnCNN.py semantics for the engine.  The loop's estimate_sigma is still
    evaluated (the reference computes it every iteration and this denoiser ignores it, F12)."""

    def __init__(self, weights, sigma):
        self.weights, self.sigma = weights, sigma

    def bind(self, batch):
        self.plan = ops.DncnnPlan(self.weights, batch.H, batch.W, batch.B)
        self.sig = torch.empty(batch.B, dtype=batch.dtype, device=batch.xrec.device)
        self.batch = batch

    def __call__(self, z, xrec, sse_out):
        from . import _native as N
        import ctypes
        b = self.batch
        N.call('pnp_sigma_est', ctypes.c_void_p(z.data_ptr()), b.H, b.W, b.B, 0 if b.dtype == torch.float32 else 1,
               ctypes.c_void_p(self.sig.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        self.plan.denoise(z, self.sigma, xrec=xrec, out=z, sse=sse_out)


class SvrgEngine:
    """pnp_svrg over a CsmriBatch.  variant='svrg' is the true direction (pnp_svrg.py:53),
    'reference' is what v1 executes (v = mu, :54).  `step(s)` = inner iteration s (the outer
    full-gradient refresh happens inside when s % T2 == 0, as in the reference's loop nest)."""

    def __init__(self, batch, prox, eta, T2, mini_batch_size, lr_decay=1.0, variant='svrg', n_log=4096, seed=0):
        self.b, self.prox, self.eta, self.T2, self.mb, self.lr_decay, self.variant = batch, prox, eta, T2, mini_batch_size, lr_decay, variant
        dev = batch.xrec.device
        self.z = batch.xinit.clone()
        self.w = torch.empty_like(self.z)
        self.mu = torch.empty_like(self.z)
        self.selT = torch.empty_like(batch.maskT)
        self.sse_log = torch.zeros((n_log, batch.B), dtype=torch.float64, device=dev)
        self.n_log = n_log
        self.seed = seed
        prox.bind(batch)
        self.s = 0
        # device-resident step counter (mirrors self.s) and scratch row: a whole outer iteration can then be
        # captured once in a hipGraph and replayed (no host-side step index inside the graph)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._dev_step = 0                                      # the value step_dev currently holds
        self.sse_tmp = torch.zeros(batch.B, dtype=torch.float64, device=dev)
        self.graph = None

    def reset(self):
        self.z.copy_(self.b.xinit)
        self.s = 0
        self.step_dev.zero_()
        self._dev_step = 0
        if hasattr(self.prox, 't'):
            self.prox.t = 0

    def step(self, idx_s=None):
        """One inner iteration for all B problems.  idx_s: int32 [B][mb] minibatch index lists (e.g. drawn
        from NumPy's legacy stream for reference-identical runs); None = draw on the device (hash keys +
        radix select, `pnp_csmri_draw_minibatch`), which keeps the draw inside the iteration like the reference."""
        b, s = self.b, self.s
        if s % self.T2 == 0:                                    # outer: mu = grad_full(z); w = z
            b.plan.grad(self.z, b.maskT, yh=b.yh_full, alpha=1.0 / float(b.M0[0]), out=self.mu)
            self.w.copy_(self.z)
        lr = self.eta * self.lr_decay ** (s // self.T2)
        if self.variant == 'svrg':
            if idx_s is None:
                b.plan.draw_minibatch(b.mask_idx, self.mb, self.seed, s, out=self.selT)
            else:
                b.plan.sel_from_indices(idx_s, out=self.selT)
            b.plan.grad(self.z, self.selT, b=self.w, alpha=-lr / self.mb, beta=1.0, c1=self.z, gamma=-lr, c2=self.mu, out=self.z)
        else:
            ops.axpbypcz(1.0, self.z, -lr, self.mu, out=self.z)
        self.prox(self.z, b.xrec, self.sse_log[s % self.n_log])
        self.s += 1                                             # eager steps keep the index on the host (no counter launch)

    # ---- hipGraph form: one OUTER iteration (full-gradient refresh + T2 inner iterations) = one graph launch
    def _outer_body(self):
        b = self.b
        b.plan.grad(self.z, b.maskT, yh=b.yh_full, alpha=1.0 / float(b.M0[0]), out=self.mu)
        self.w.copy_(self.z)
        lr = self.eta
        for _ in range(self.T2):
            if self.variant == 'svrg':
                b.plan.draw_minibatch(b.mask_idx, self.mb, self.seed, 0, out=self.selT, step_dev=self.step_dev)
                b.plan.grad(self.z, self.selT, b=self.w, alpha=-lr / self.mb, beta=1.0, c1=self.z, gamma=-lr, c2=self.mu, out=self.z)
            else:
                ops.axpbypcz(1.0, self.z, -lr, self.mu, out=self.z)
            self.prox(self.z, b.xrec, self.sse_tmp)
            ops.log_append(self.sse_tmp, self.sse_log, self.step_dev)
            ops.counter_add(self.step_dev, 1)

    def capture(self):
        """Capture one outer iteration into a hipGraph (torch.cuda.CUDAGraph on ROCm).  Needs lr_decay == 1, a
        step count that is a multiple of T2, device-side minibatch draws and a prox without host-side
        per-call state (TVProx with denoise_strength == 0, DnCNNProx).  State is left untouched."""
        assert self.lr_decay == 1.0 and self.s % self.T2 == 0
        assert getattr(self.prox, 'denoise_strength', 0.0) == 0.0
        self._set_dev_step(self.s)
        keep = (self.z.clone(), self.w.clone(), self.mu.clone(), self.sse_log.clone(), self.step_dev.clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._outer_body()                                  # warm-up outside capture (lazy module loads)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._outer_body()
        torch.cuda.synchronize()
        for dst, src in zip((self.z, self.w, self.mu, self.sse_log, self.step_dev), keep):
            dst.copy_(src)
        self.graph = g
        return g

    def _set_dev_step(self, s):
        """The device-resident counter is brought up to date only when a graph is about to read it."""
        if self._dev_step != s:
            self.step_dev.fill_(s)
            self._dev_step = s

    def run_outer(self, n_outer=1):
        """n_outer graph replays = n_outer * T2 inner iterations."""
        if self.graph is None:
            self.capture()
        self._set_dev_step(self.s)
        for _ in range(n_outer):
            self.graph.replay()
            self.s += self.T2
        self._dev_step = self.s

    def psnr_trace(self):
        """[steps][B] PSNR (rounded to 0.01 dB like problems/problem.py:33-35), read back once."""
        sse = self.sse_log[:min(self.s, self.n_log)].cpu().numpy()
        with np.errstate(divide='ignore'):
            return np.around(10 * np.log10(1.0 / (sse / self.b.N)), 2)


class SvrgEngineFusedTV(SvrgEngine):
    """SvrgEngine for the TV prox with the inner iteration's last three stages in ONE kernel
    (`pnp_csmri_grad_prox_tv`: SVRG step -> estimate_sigma -> Haar BayesShrink -> PSNR error).

    The prox works along image columns, the last pass of the inverse FFT hands out storage rows; so this engine
    keeps every image TRANSPOSED in HBM (z, w, mu, ground truth) and feeds the plan the transposed problem
    (fft2(x^T) = fft2(x)^T: the un-transposed mask and data where the plain engine passes transposed ones).  `z`
    reads back un-transposed.  Same minibatches, same results as SvrgEngine up to the summation order of the
    wavelet sub-band energies (~1e-6); the stepped image never travels to HBM between gradient and prox.
    Opt-in (`make_engine(..., fused=True)`): on MI355X the one-workgroup-per-image kernel is measured SLOWER than
    the two streaming kernels it replaces (csmri.hip, k_rows_inv_prox), so the plain engine stays the default."""

    @staticmethod
    def eligible(batch, prox, variant='svrg'):
        return (isinstance(prox, TVProx) and variant == 'svrg' and batch.dtype == torch.float32
                and batch.H == batch.W and batch.H in (64, 256))

    def __init__(self, batch, prox, eta, T2, mini_batch_size, lr_decay=1.0, variant='svrg', n_log=4096, seed=0):
        assert self.eligible(batch, prox, variant), 'fused TV engine: TVProx, true SVRG, float32, 64x64 or 256x256'
        self.b, self.prox, self.eta, self.T2, self.mb, self.lr_decay, self.variant = batch, prox, eta, T2, mini_batch_size, lr_decay, variant
        dev = batch.xrec.device
        tr = lambda a: a.transpose(1, 2).contiguous()
        self._zT = tr(batch.xinit)
        self._wT = torch.empty_like(self._zT)
        self._muT = torch.empty_like(self._zT)
        self._xrecT = tr(batch.xrec)
        # the transposed problem: its "transposed selector" is the plain mask, its data term is packed from Y itself
        self._mask = torch.from_numpy(batch.mask_np).to(dev)
        cdt = torch.complex64
        self._yh_full = batch.plan.pack_y(batch.Y_dev(cdt), self._mask)
        self._mask_idxT = self._tidx(batch.mask_idx)            # same order as mask_idx -> the same draws
        self.selT = torch.empty_like(self._mask)
        self.sse_log = torch.zeros((n_log, batch.B), dtype=torch.float64, device=dev)
        self.n_log, self.seed, self.s = n_log, seed, 0
        prox.bind(batch)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._dev_step = 0
        self.sse_tmp = torch.zeros(batch.B, dtype=torch.float64, device=dev)
        self.graph = None

    def _tidx(self, idx):
        """flat row-major index into H x W -> flat index of the same location in the transposed W x H array"""
        H, W = self.b.H, self.b.W
        return ((idx % W) * H + idx // W).to(torch.int32)

    z = property(lambda self: self._zT.transpose(1, 2))
    w = property(lambda self: self._wT.transpose(1, 2))
    mu = property(lambda self: self._muT.transpose(1, 2))

    def _refresh(self):
        b = self.b
        b.plan.grad(self._zT, self._mask, yh=self._yh_full, alpha=1.0 / float(b.M0[0]), out=self._muT)
        self._wT.copy_(self._zT)

    def _inner(self, lr, sse_out):
        b, px = self.b, self.prox
        px.t += 1
        b.plan.grad_prox_tv(self._zT, self.selT, b=self._wT, alpha=-lr / self.mb, beta=1.0, c1=self._zT, gamma=-lr,
                            c2=self._muT, out=self._zT, sigma_modifier=px.sigma_modifier,
                            fallback_sigma=px.denoise_strength * px.decay ** px.t, xrec=self._xrecT, sse=sse_out,
                            sigma_out=px.sig)

    def step(self, idx_s=None):
        b, s = self.b, self.s
        if s % self.T2 == 0:
            self._refresh()
        lr = self.eta * self.lr_decay ** (s // self.T2)
        if idx_s is None:
            b.plan.draw_minibatch(self._mask_idxT, self.mb, self.seed, s, out=self.selT)
        else:
            b.plan.sel_from_indices(self._tidx(idx_s), out=self.selT)
        self._inner(lr, self.sse_log[s % self.n_log])
        self.s += 1

    def _outer_body(self):
        b = self.b
        self._refresh()
        for _ in range(self.T2):
            b.plan.draw_minibatch(self._mask_idxT, self.mb, self.seed, 0, out=self.selT, step_dev=self.step_dev)
            self._inner(self.eta, self.sse_tmp)
            ops.log_append(self.sse_tmp, self.sse_log, self.step_dev)
            ops.counter_add(self.step_dev, 1)


def make_engine(batch, prox, eta, T2, mini_batch_size, lr_decay=1.0, variant='svrg', fused=None, **kw):
    """SvrgEngine (default: the faster form, see SvrgEngineFusedTV), or SvrgEngineFusedTV when fused=True."""
    ok = SvrgEngineFusedTV.eligible(batch, prox, variant)
    if fused and not ok:
        raise ValueError('fused TV engine needs TVProx, variant="svrg", float32 and 64x64 or 256x256 images')
    cls = SvrgEngineFusedTV if fused else SvrgEngine
    return cls(batch, prox, eta, T2, mini_batch_size, lr_decay=lr_decay, variant=variant, **kw)
