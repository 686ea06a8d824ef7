"""Batched, sync-free PnP engines: B independent reconstructions advance together, one inner iteration per
`step()`, entirely in HBM.

These are the throughput forms of the reference loops -- algorithms/pnp_svrg.py:26-95, pnp_sgd.py:24-70,
pnp_gd.py:24-70, pnp_sarah.py:28-104, pnp_saga.py:25-72 -- for the sweep-style workloads of script_diff_*_set12.py,
where many reconstructions (image x sampling ratio x SNR x trial) are independent.  Nothing is read back per
iteration: squared errors (for PSNR) accumulate in a device log.  The drop-in loops of `algorithms.py` (golden-pinned)
are the specification: every engine is tested to walk the same trajectory as B drop-in loops fed the same minibatches.

A *batch problem* (CsmriBatch, DeblurBatch, PrBatch) holds the device-resident data of B problems and offers
    grad_full(z, out, alpha, beta, c1)                         alpha * grad_full(z) + beta * c1
    grad_stoch(z, mbs, j, out, alpha, beta, c1)                alpha * grad_stoch(z, minibatch j) + beta * c1
    grad_stoch_diff(z, w, mbs, j, out, alpha, beta, c1, gamma, c2)
                                                               alpha * (gs(z) - gs(w)) + beta * c1 + gamma * c2
    minibatches(n) / draw(mbs, mb, seed, step0, nsteps) / set_host(mbs, j, idx)
Minibatches are drawn on the device by default (counter-based keys + a threshold per (problem, step): csrc/draw.h;
the selection itself is re-derived inside the gradient kernels and never stored); for reference-identical runs pass
index lists drawn from the legacy `np.random` stream (`step(idx)`).
"""
import numpy as np
import torch

from . import ops


class Minibatches:
    """n slots of per-problem minibatch selections: threshold descriptors of device draws (int64 [n, B, 2]) or
    host-provided selections (slot -> whatever the batch problem's kernels take)."""

    def __init__(self, n, B, device, bits_shape=None):
        self.n = n
        self.mbd = torch.zeros((n, B, 2), dtype=torch.int64, device=device)
        # CSMRI: the device-drawn selections themselves, bit-packed (what the column pass reads)
        self.selbits = torch.zeros((n, B) + tuple(bits_shape), dtype=torch.int32, device=device) if bits_shape else None
        self.host = [None] * n


class _BatchBase:
    def minibatches(self, n):
        return Minibatches(n, self.B, self.device)

    def _check_mb(self, mb):
        if mb > self.max_mb:
            # np.random.choice(..., replace=False) raises the same way (problems/problem.py:110-117, CSMRI.py:66-74)
            raise ValueError(f"Cannot take a larger sample than population when 'replace=False' (mini_batch_size {mb} > {self.max_mb})")

    def psnr_init(self):
        """rounded PSNR of Xinit per problem (problems/problem.py:33-35)."""
        sse = ops.sse(self.xinit, self.xrec).cpu().numpy()
        with np.errstate(divide='ignore'):
            return np.around(10 * np.log10(1.0 / (sse / self.N)), 2)


class CsmriBatch(_BatchBase):
    """Device-resident data of B CSMRI problems (reference problems/CSMRI.py:12-41 per problem).  Masks may have
    different numbers of sampled locations (the reference draws Bernoulli masks, CSMRI.py:43-45): grad_full's 1/M0
    is a per-problem device vector."""
    kind = 'csmri'

    def __init__(self, xrec, mask, Y, xinit, dtype=torch.float32, device='cuda'):
        ops.require_gpu()
        B, H, W = xrec.shape
        self.B, self.H, self.W, self.N, self.dtype = B, H, W, H * W, dtype
        self.device = torch.device(device)
        cdt = torch.complex64 if dtype == torch.float32 else torch.complex128
        self.plan = ops.CsmriPlan(H, W, B, dtype)
        self.xrec = torch.from_numpy(np.ascontiguousarray(xrec, np.float64)).to(device, dtype)
        self.xinit = torch.from_numpy(np.ascontiguousarray(xinit, np.float64)).to(device, dtype).reshape(B, H, W)
        self.mask_np = np.ascontiguousarray(mask, np.uint8).reshape(B, H, W)
        self.M0 = self.mask_np.reshape(B, -1).sum(1).astype(np.int64)
        self.max_mb = int(self.M0.min())
        self.inv_m0 = torch.from_numpy(1.0 / self.M0.astype(np.float64)).to(device, dtype)
        self.maskT = self.plan.sel_from_dense(torch.from_numpy(self.mask_np).to(device))
        self.bits = self.plan.pack_mask(self.maskT)
        self.YT = torch.from_numpy(np.ascontiguousarray(np.swapaxes(Y, 1, 2))).to(device, cdt).contiguous()
        self.yh_full = self.plan.pack_y(self.YT, self.maskT)

    @classmethod
    def synthetic(cls, B, H=256, W=256, sample_prob=0.2, snr=20.0, seed=0, dtype=torch.float32, bernoulli=True):
        """B synthetic problems: smoothed-noise images, complex data with real noise on the support.
        bernoulli=True draws each mask entry with probability p like the reference (CSMRI.py:43-45; the number of
        sampled points then differs per problem), False draws exactly round(p*N) points."""
        rng = np.random.default_rng(seed)
        N = H * W
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
            if bernoulli:
                mask[b] = rng.random(N) < sample_prob
            else:
                mask[b, rng.choice(N, int(round(sample_prob * N)), replace=False)] = 1
            mk = mask[b].reshape(H, W)
            Y0 = mk * np.fft.fft2(xrec[b])
            sigma = np.sqrt(np.linalg.norm(Y0.ravel()) / 10 ** (snr / 10) / H / W)     # problem.py:58-61
            Y[b] = Y0 + mk * rng.normal(0, sigma, (H, W))
            xi = np.absolute(np.fft.ifft2(Y[b])).ravel()
            xinit[b] = (xi - xi.min()) / (xi.max() - xi.min())
        return cls(xrec, mask.reshape(B, H, W), Y, xinit, dtype=dtype)

    @classmethod
    def from_problems(cls, probs, dtype=torch.float32, device='cuda'):
        """From reference-style problem objects (anything with Xrec, mask, Y, Xinit: problems.CSMRI, the oracle's)."""
        return cls(np.stack([p.Xrec for p in probs]), np.stack([p.mask for p in probs]), np.stack([p.Y for p in probs]),
                   np.stack([p.Xinit for p in probs]), dtype=dtype, device=device)

    def draw_minibatches(self, n_steps, mb, seed=1):
        """[n_steps][B][mb] int32 flat k-space indices, each row a uniform draw without replacement
        from that problem's mask support (CSMRI.py:66-74 semantics, fast Generator stream)."""
        self._check_mb(mb)
        rng = np.random.default_rng(seed)
        out = np.empty((n_steps, self.B, mb), np.int32)
        for b in range(self.B):
            locs = np.flatnonzero(self.mask_np[b]).astype(np.int32)
            for s in range(n_steps):
                out[s, b] = rng.choice(locs, mb, replace=False)
        return torch.from_numpy(out).to(self.device)

    # ---- minibatch slots
    def minibatches(self, n):
        return Minibatches(n, self.B, self.device, bits_shape=(self.W, self.H // 32))

    def draw(self, mbs, mb, seed, step0, nsteps=1, step_dev=None):
        self._check_mb(mb)
        self.plan.draw_thresholds(self.bits, mb, seed, step0, nsteps, out=mbs.mbd[:nsteps], step_dev=step_dev,
                                  selbits=mbs.selbits[:nsteps])
        for j in range(nsteps):
            mbs.host[j] = None

    def set_host(self, mbs, j, idx):
        """idx: int32 [B, mb] flat row-major k-space positions (np.flatnonzero(mask o minibatch))."""
        mbs.host[j] = self.plan.sel_from_indices(idx, out=mbs.host[j] if isinstance(mbs.host[j], torch.Tensor) else None)

    def _sel(self, mbs, j):
        if mbs.host[j] is not None:
            return dict(selT=mbs.host[j])
        return dict(bits=mbs.selbits[j])

    # ---- gradients
    def grad_full(self, z, out, alpha=1.0, beta=0.0, c1=None):
        return self.plan.grad(z, bits=self.bits, yh=self.yh_full, alpha=alpha, alpha_vec=self.inv_m0, beta=beta, c1=c1, out=out)

    def grad_stoch(self, z, mbs, j, out, alpha=1.0, beta=0.0, c1=None):
        return self.plan.grad(z, YT=self.YT, alpha=alpha, beta=beta, c1=c1, out=out, **self._sel(mbs, j))

    def grad_stoch_diff(self, z, w, mbs, j, out, alpha=1.0, beta=0.0, c1=None, gamma=0.0, c2=None):
        # one FFT pair: the data terms cancel
        return self.plan.grad(z, b=w, alpha=alpha, beta=beta, c1=c1, gamma=gamma, c2=c2, out=out, **self._sel(mbs, j))


class DeblurBatch(_BatchBase):
    """B Deblur / super-resolution problems sharing one blur kernel and one down-sampler (reference
    problems/DeblurSR.py:17-147 per problem; the sweeps vary image, noise and seed, not the operator)."""
    kind = 'deblur'

    def __init__(self, xrec, Bk, Y, xinit, dtype=torch.float32, device='cuda', bilinear=None):
        ops.require_gpu()
        B, H, W = xrec.shape
        self.B, self.H, self.W, self.N, self.dtype = B, H, W, H * W, dtype
        self.device = torch.device(device)
        self.plan = ops.DeblurPlan(H, W, B, dtype, Bk, bilinear=bilinear)
        self.M = self.plan.M
        self.max_mb = self.M
        self.xrec = torch.from_numpy(np.ascontiguousarray(xrec, np.float64)).to(device, dtype)
        self.xinit = torch.from_numpy(np.ascontiguousarray(xinit, np.float64)).to(device, dtype).reshape(B, H, W)
        self.Y = torch.from_numpy(np.ascontiguousarray(Y, np.float64)).to(device, dtype).reshape(B, self.M)
        self._tmp = None

    @classmethod
    def from_problems(cls, probs, dtype=torch.float32, device='cuda'):
        p0 = probs[0]
        return cls(np.stack([p.Xrec for p in probs]), p0.B, np.stack([p.Y for p in probs]),
                   np.stack([p.Xinit for p in probs]), dtype=dtype, device=device,
                   bilinear=getattr(p0, 'Bop', None) if isinstance(getattr(p0, 'Bop', None), tuple) else None)

    @classmethod
    def synthetic(cls, B, H=256, W=256, kernel='Minimal', snr=20.0, seed=0, dtype=torch.float32):
        """B synthetic Deblur problems (scale_percent = 100): smoothed-noise images, the reference's "Minimal" or
        "Identity" kernel (DeblurSR.py:80-89), noise and U(0,1) initialisation from a Generator stream."""
        rng = np.random.default_rng(seed)
        N = H * W
        if kernel == 'Minimal':
            Bk = np.zeros((H, W))
            Bk[0, 0] = Bk[H // 2, H // 2] = Bk[H // 2, H // 3] = Bk[H // 2, H // 4] = 0.25
        else:
            Bk = np.zeros((H, W))
            Bk[0, 0] = 1
        Bk = Bk.ravel() / N
        FB = np.fft.fft(Bk)
        xrec = np.empty((B, H, W))
        Y = np.empty((B, N))
        for b in range(B):
            x = rng.random((H, W))
            p = np.pad(x, 2, mode='wrap')
            y = sum(p[i:i + H, j:j + W] for i in range(5) for j in range(5)) / 25.0
            xrec[b] = (y - y.min()) / (y.max() - y.min())
            Y0 = np.real(np.fft.ifft(np.fft.fft(xrec[b].ravel()) * FB)) * np.sqrt(N)       # DeblurSR.py:119-120
            sigma = np.sqrt(np.linalg.norm(Y0) / 10 ** (snr / 10) / H / W)
            Y[b] = Y0 + rng.normal(0, sigma, N)
        return cls(xrec, Bk, Y, rng.uniform(0.0, 1.0, (B, N)), dtype=dtype)

    def draw_minibatches(self, n_steps, mb, seed=1):
        self._check_mb(mb)
        rng = np.random.default_rng(seed)
        out = np.stack([[rng.choice(self.M, mb, replace=False) for _ in range(self.B)] for _ in range(n_steps)]).astype(np.int32)
        return torch.from_numpy(out).to(self.device)

    def draw(self, mbs, mb, seed, step0, nsteps=1, step_dev=None):
        self._check_mb(mb)
        ops.draw_thresholds(self.M, self.B, mb, seed, step0, nsteps, out=mbs.mbd[:nsteps], step_dev=step_dev)
        for j in range(nsteps):
            mbs.host[j] = None

    def set_host(self, mbs, j, idx):
        """idx: int32 [B, mb] measurement indices (np.flatnonzero of Problem.select_mb's indicator)."""
        mbs.host[j] = ops.indicator_from_indices(idx, self.M, out=mbs.host[j] if isinstance(mbs.host[j], torch.Tensor) else None)

    def _sel(self, mbs, j):
        if mbs.host[j] is not None:
            return dict(sel=mbs.host[j])
        return dict(mbd=mbs.mbd[j])

    def _axpby(self, g, alpha_applied, out, beta, c1, gamma=0.0, c2=None):
        if c1 is None and c2 is None:
            return g
        return ops.axpbypcz(1.0, g, beta, c1, gamma, c2, out=out)

    def _scratch(self, z):
        if self._tmp is None:
            self._tmp = torch.empty_like(z)
        return self._tmp

    def grad_full(self, z, out, alpha=1.0, beta=0.0, c1=None):
        if c1 is None:
            return self.plan.grad(z, self.Y, scale=alpha / self.M, out=out)
        g = self.plan.grad(z, self.Y, scale=alpha / self.M, out=self._scratch(z))
        return ops.axpbypcz(1.0, g, beta, c1, out=out)

    def grad_stoch(self, z, mbs, j, out, alpha=1.0, beta=0.0, c1=None):
        if c1 is None:
            return self.plan.grad(z, self.Y, scale=alpha, out=out, **self._sel(mbs, j))
        g = self.plan.grad(z, self.Y, scale=alpha, out=self._scratch(z), **self._sel(mbs, j))
        return ops.axpbypcz(1.0, g, beta, c1, out=out)

    def grad_stoch_diff(self, z, w, mbs, j, out, alpha=1.0, beta=0.0, c1=None, gamma=0.0, c2=None):
        # gs(z) - gs(w) = B^T S^T sel (S B (z - w)) + (terms in y cancel): two gradients, one combine
        t = self._scratch(z)
        g1 = self.plan.grad(z, self.Y, scale=alpha, out=torch.empty_like(z), **self._sel(mbs, j))
        g2 = self.plan.grad(w, self.Y, scale=alpha, out=t, **self._sel(mbs, j))
        d = ops.axpbypcz(1.0, g1, -1.0, g2, out=g1)
        if c1 is None and c2 is None:
            return out.copy_(d) if out is not d else d
        return ops.axpbypcz(1.0, d, beta, c1, gamma, c2, out=out)


class PrBatch(_BatchBase):
    """B phase-retrieval problems (reference problems/PR.py:13-87 per problem), each with its own dense M x N matrix."""
    kind = 'pr'

    def __init__(self, xrec, A, Y, xinit, dtype=torch.float32, device='cuda'):
        ops.require_gpu()
        B, H, W = xrec.shape
        self.B, self.H, self.W, self.N, self.dtype = B, H, W, H * W, dtype
        self.device = torch.device(device)
        self.M = A.shape[1]
        self.max_mb = self.M
        self.xrec = torch.from_numpy(np.ascontiguousarray(xrec, np.float64)).to(device, dtype)
        self.xinit = torch.from_numpy(np.ascontiguousarray(xinit, np.float64)).to(device, dtype).reshape(B, H, W)
        self.A = torch.from_numpy(np.ascontiguousarray(A, np.float64)).to(device, dtype).contiguous()
        self.Y = torch.from_numpy(np.ascontiguousarray(Y, np.float64)).to(device, dtype).reshape(B, self.M)
        from . import _native as N
        self._ws = torch.empty(B * N.lib().pnp_pr_workspace_elems(self.M, self.N), dtype=dtype, device=device)
        self._tmp = None
        self._mb = None

    @classmethod
    def from_problems(cls, probs, dtype=torch.float32, device='cuda'):
        return cls(np.stack([p.Xrec for p in probs]), np.stack([p.A for p in probs]), np.stack([p.Y for p in probs]),
                   np.stack([p.Xinit for p in probs]), dtype=dtype, device=device)

    def draw_minibatches(self, n_steps, mb, seed=1):
        self._check_mb(mb)
        rng = np.random.default_rng(seed)
        out = np.stack([[np.sort(rng.choice(self.M, mb, replace=False)) for _ in range(self.B)] for _ in range(n_steps)]).astype(np.int32)
        return torch.from_numpy(out).to(self.device)

    def draw(self, mbs, mb, seed, step0, nsteps=1, step_dev=None):
        self._check_mb(mb)
        self._mb = mb
        ops.draw_thresholds(self.M, self.B, mb, seed, step0, nsteps, out=mbs.mbd[:nsteps], step_dev=step_dev)
        for j in range(nsteps):
            mbs.host[j] = None

    def set_host(self, mbs, j, idx):
        """idx: int32 [B, mb] row ids (np.flatnonzero of the indicator: ascending, like A[idx] in PR.py:82-83)."""
        mbs.host[j] = idx.contiguous()

    def _rows(self, mbs, j):
        if mbs.host[j] is not None:
            return mbs.host[j]
        key = ('rows', j)
        buf = getattr(mbs, '_rows', None)
        if buf is None:
            buf = mbs._rows = {}
        if key not in buf:
            buf[key] = torch.empty((self.B, self._mb), dtype=torch.int32, device=self.device)
        return ops.rows_from_thresholds(self.M, self._mb, mbs.mbd[j], out=buf[key])

    def _scratch(self, z):
        if self._tmp is None:
            self._tmp = torch.empty((self.B, self.N), dtype=self.dtype, device=self.device)
        return self._tmp

    def _g(self, z, rows, scale, out):
        return ops.pr_grad_batch(self.A, z.reshape(self.B, self.N), self.Y, rows=rows, scale=scale, workspace=self._ws,
                                 out=out.reshape(self.B, self.N)).reshape(out.shape)

    def grad_full(self, z, out, alpha=1.0, beta=0.0, c1=None):
        if c1 is None:
            return self._g(z, None, alpha / self.M, out)
        g = self._g(z, None, alpha / self.M, self._scratch(z))
        return ops.axpbypcz(1.0, g.reshape(z.shape), beta, c1, out=out)

    def grad_stoch(self, z, mbs, j, out, alpha=1.0, beta=0.0, c1=None):
        rows = self._rows(mbs, j)
        if c1 is None:
            return self._g(z, rows, alpha, out)
        g = self._g(z, rows, alpha, self._scratch(z))
        return ops.axpbypcz(1.0, g.reshape(z.shape), beta, c1, out=out)

    def grad_stoch_diff(self, z, w, mbs, j, out, alpha=1.0, beta=0.0, c1=None, gamma=0.0, c2=None):
        rows = self._rows(mbs, j)
        g1 = self._g(z, rows, alpha, torch.empty_like(z))
        g2 = self._g(w, rows, alpha, self._scratch(z)).reshape(z.shape)
        d = ops.axpbypcz(1.0, g1, -1.0, g2, out=g1)
        if c1 is None and c2 is None:
            return out.copy_(d) if out is not d else d
        return ops.axpbypcz(1.0, d, beta, c1, gamma, c2, out=out)


# ---------------------------------------------------------------------------------------------------------- prox
class TVProx:
    """denoisers/TV.py semantics for the engines (fused estimate_sigma + BayesShrink + error sum)."""

    def __init__(self, sigma_modifier=1.0, decay=1.0, denoise_strength=0.0):
        self.sigma_modifier, self.decay, self.denoise_strength, self.t = sigma_modifier, decay, denoise_strength, 0

    def bind(self, batch):
        self.sig = torch.empty(batch.B, dtype=batch.dtype, device=batch.xrec.device)

    def __call__(self, z, xrec, sse_out):
        self.t += 1
        ops.prox_tv(z, sigma_modifier=self.sigma_modifier, fallback_sigma=self.denoise_strength * self.decay ** self.t,
                    xrec=xrec, out=z, sse=sse_out, sigma_out=self.sig)
        return z

    inplace = True                                              # writes its result into the iterate it was given
    # one-kernel iteration (pnp_csmri_svrg_step): the prox runs inside the gradient kernel
    fused_denoise = True

    def fused_args(self):
        self.t += 1
        return dict(sigma_modifier=self.sigma_modifier, fallback_sigma=self.denoise_strength * self.decay ** self.t, sigma_out=self.sig)

    def after_fused(self, z, xrec, sse_out):
        return z


class DnCNNProx:
    """denoisers/RealSN_DnCNN.py semantics for the engines.  The loop's estimate_sigma is still
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
        return z

    inplace = True
    # one-kernel iteration: the gradient kernel makes the (ignored, F12) noise estimate; the network follows
    fused_denoise = False

    def fused_args(self):
        return dict(sigma_out=self.sig)

    def after_fused(self, z, xrec, sse_out):
        self.plan.denoise(z, self.sigma, xrec=xrec, out=z, sse=sse_out)
        return z


class NLMProx:
    """denoisers/NLM.py:22-27 semantics for the engines: h = sigma = estimate_sigma * sigma_modifier when
    `self.sigma > 0` (the attribute the reference reads; default 1.0 here), else the decaying fixed strength.
    NLM cannot run in place: the prox ping-pongs between the engine's iterate and a buffer of its own and RETURNS the
    tensor that holds the result."""

    inplace = False                                             # ping-pongs: a hipGraph of an outer iteration cannot hold it

    def __init__(self, sigma=1.0, sigma_modifier=1.0, decay=1.0, denoise_strength=0.0, patch_size=4, patch_distance=5):
        self.sigma, self.sigma_modifier, self.decay, self.denoise_strength = sigma, sigma_modifier, decay, denoise_strength
        self.patch_size, self.patch_distance, self.t = patch_size, patch_distance, 0

    def bind(self, batch):
        self.sig = torch.empty(batch.B, dtype=batch.dtype, device=batch.xrec.device)
        self.buf = torch.empty_like(batch.xinit)

    def __call__(self, z, xrec, sse_out):
        self.t += 1
        if self.sigma > 0:
            from . import _native as N
            import ctypes
            B, H, W = z.shape
            N.call('pnp_sigma_est', ctypes.c_void_p(z.data_ptr()), H, W, B, 0 if z.dtype == torch.float32 else 1,
                   ctypes.c_void_p(self.sig.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            ops.nlm2d(z, sigma_in=self.sig, sigma_modifier=self.sigma_modifier, patch_size=self.patch_size,
                      patch_distance=self.patch_distance, xrec=xrec, out=self.buf, sse=sse_out)
        else:
            ops.nlm2d(z, fixed_h=self.denoise_strength * self.decay ** self.t, patch_size=self.patch_size,
                      patch_distance=self.patch_distance, xrec=xrec, out=self.buf, sse=sse_out)
        out, self.buf = self.buf, z
        return out


# ---------------------------------------------------------------------------------------------------------- engines
class LoopEngine:
    """State and log machinery shared by the engines: the iterate z [B, H, W], a device log ring of the squared errors
    of every prox evaluation (-> rounded PSNR traces like the reference's psnr_per_iter), the step counter."""

    def __init__(self, batch, prox, eta, lr_decay=1.0, n_log=4096, seed=0):
        self.b, self.prox, self.eta, self.lr_decay, self.seed = batch, prox, eta, lr_decay, seed
        dev = batch.xrec.device
        self.z = batch.xinit.clone()
        self.sse_log = torch.zeros((n_log, batch.B), dtype=torch.float64, device=dev)
        self.n_log = n_log
        prox.bind(batch)
        self.s = 0                                              # inner iterations done
        self.n_prox = 0                                         # prox evaluations logged
        self.graph = None

    def reset(self):
        self.z.copy_(self.b.xinit)
        self.s = self.n_prox = 0
        if hasattr(self.prox, 't'):
            self.prox.t = 0

    def _prox(self, z):
        out = self.prox(z, self.b.xrec, self.sse_log[self.n_prox % self.n_log])
        self.n_prox += 1
        return out

    def psnr_trace(self):
        """[prox evaluations][B] PSNR (rounded to 0.01 dB like problems/problem.py:33-35) in chronological order,
        read back once; when more than n_log evaluations were logged, the last n_log of them."""
        n = min(self.n_prox, self.n_log)
        log = self.sse_log[:n]
        if self.n_prox > self.n_log:
            log = torch.roll(self.sse_log, -(self.n_prox % self.n_log), 0)     # oldest surviving row first
        sse = log.cpu().numpy()
        with np.errstate(divide='ignore'):
            return np.around(10 * np.log10(1.0 / (sse / self.b.N)), 2)


class GdEngine(LoopEngine):
    """pnp_gd over a batch (algorithms/pnp_gd.py:24-70): z <- prox(z - eta * decay^i * grad_full(z))."""

    def step(self):
        lr = self.eta * self.lr_decay ** self.s
        self.b.grad_full(self.z, out=self.z, alpha=-lr, beta=1.0, c1=self.z)
        self.z = self._prox(self.z)
        self.s += 1


class _StochEngine(LoopEngine):
    # Engines that draw one minibatch per step (sgd, saga, sarah) draw AHEAD steps per launch: the draw kernel is
    # latency-bound (a 32-step radix select per problem, ~50 us whatever the grid), and a step's descriptor depends on
    # (seed, step, problem) only, so a window of steps drawn together holds the very same selections.
    AHEAD = 16

    def __init__(self, batch, prox, eta, mini_batch_size, lr_decay=1.0, n_log=4096, seed=0, n_slots=None):
        super().__init__(batch, prox, eta, lr_decay, n_log, seed)
        batch._check_mb(mini_batch_size)
        self.mb = mini_batch_size
        self._window = n_slots is None                          # one draw launch per AHEAD steps (else: the subclass draws)
        self.mbs = batch.minibatches(self.AHEAD if n_slots is None else n_slots)
        self._drawn_base = None                                 # first step of the window the slots currently hold

    def _minibatch(self, idx_s, step_id):
        """Bind a slot to this step's minibatch and return it: host index lists when given (slot 0), else a device draw."""
        if idx_s is not None:
            self.b.set_host(self.mbs, 0, idx_s)
            self._drawn_base = None
            return 0
        if not self._window or step_id >= 0xFFFFFFF0:          # (0xFFFFFFFF: the table-filling draw of pnp_saga)
            self.b.draw(self.mbs, self.mb, self.seed, step_id, 1)
            self._drawn_base = None
            return 0
        base = step_id - step_id % self.AHEAD
        if self._drawn_base != base:
            self.b.draw(self.mbs, self.mb, self.seed, base, self.AHEAD)
            self._drawn_base = base
        return step_id - base

    def _draw_slot(self, slot, step_id):
        one = Minibatches.__new__(Minibatches)
        one.n, one.mbd, one.host = 1, self.mbs.mbd[slot:slot + 1], [None]
        one.selbits = self.mbs.selbits[slot:slot + 1] if self.mbs.selbits is not None else None
        self.b.draw(one, self.mb, self.seed, step_id, 1)
        self.mbs.host[slot] = None


class SgdEngine(_StochEngine):
    """pnp_sgd over a batch (algorithms/pnp_sgd.py:24-70): v = grad_stoch(z, mb) / mini_batch_size."""

    def step(self, idx_s=None):
        j = self._minibatch(idx_s, self.s)
        lr = self.eta * self.lr_decay ** self.s
        self.b.grad_stoch(self.z, self.mbs, j, out=self.z, alpha=-lr / self.mb, beta=1.0, c1=self.z)
        self.z = self._prox(self.z)
        self.s += 1


class SvrgEngine(_StochEngine):
    """pnp_svrg over a batch.  variant='svrg' is the true direction (pnp_svrg.py:53), 'reference' is what v1
    executes (v = mu, :54).  `step()` = one inner iteration (the outer full-gradient refresh happens inside when
    s % T2 == 0, as in the reference's loop nest).  Device draws of a whole outer iteration are ONE launch at the
    refresh (T2 descriptor slots)."""
    FUSED_MIN_BATCH = 192

    def __init__(self, batch, prox, eta, T2, mini_batch_size, lr_decay=1.0, variant='svrg', n_log=4096, seed=0, fused=None,
                 fold_outer=True):
        super().__init__(batch, prox, eta, mini_batch_size, lr_decay, n_log, seed, n_slots=T2)
        self.T2, self.variant = T2, variant
        # the one-kernel inner iteration (csrc/csmri_fused.hip): CSMRI, f32, 256 x 256, true SVRG direction, a prox that
        # can follow it (TV inside the kernel, DnCNN after it)
        ok = (batch.kind == 'csmri' and batch.dtype == torch.float32 and batch.H == 256 and batch.W == 256
              and variant == 'svrg' and hasattr(prox, 'fused_args'))
        if fused and not ok:
            raise ValueError('the one-kernel iteration needs a float32 256 x 256 CsmriBatch, variant="svrg" and a TV or DnCNN prox')
        # one workgroup per image: it pays off from about a workgroup per CU (measured: B = 120 is slower than the
        # streaming kernels, B = 256 faster), so small batches keep the four streaming kernels unless asked otherwise
        self.fused = (ok and batch.B >= self.FUSED_MIN_BATCH) if fused is None else bool(fused)
        self.fold_outer = fold_outer                            # False: refresh as launches of its own (A/B, tests)
        self._hostbits = None
        dev = batch.xrec.device
        self.w = torch.empty_like(self.z)
        self.mu = torch.empty_like(self.z)
        # device-resident step counter (mirrors self.s) and scratch row: a whole outer iteration can then be
        # captured once in a hipGraph and replayed (no host-side step index inside the graph)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._dev_step = 0                                      # the value step_dev currently holds
        self.sse_tmp = torch.zeros(batch.B, dtype=torch.float64, device=dev)

    def reset(self):
        super().reset()
        self.step_dev.zero_()
        self._dev_step = 0

    def step(self, idx_s=None):
        """One inner iteration for all B problems.  idx_s: int32 [B][mb] minibatch index lists (e.g. drawn from
        NumPy's legacy stream for reference-identical runs); None = device draws."""
        b, s = self.b, self.s
        j = s % self.T2
        lr = self.eta * self.lr_decay ** (s // self.T2)
        if j == 0:                                              # outer: mu = grad_full(z); w = z
            if self.variant == 'svrg' and idx_s is None:
                b.draw(self.mbs, self.mb, self.seed, s, self.T2)
            if self.fused and self.fold_outer:
                # ... folded into the first inner iteration: at j = 0 the SVRG difference gs(z) - gs(w) is exactly zero
                # (w == z), so that iteration is z <- prox(z - lr * mu); ONE kernel forms mu, stores it and w, and goes on
                if idx_s is not None:                           # (the minibatch of this step is drawn but cannot matter)
                    b.set_host(self.mbs, j, idx_s)
                self._fused_outer(lr, self.sse_log[self.n_prox % self.n_log])
                self.n_prox += 1
                self.s += 1
                return
            b.grad_full(self.z, out=self.mu)
            self.w.copy_(self.z)
        if self.variant == 'svrg':
            if idx_s is not None:
                b.set_host(self.mbs, j, idx_s)
            elif self.mbs.host[j] is not None:                  # a host-fed outer iteration continued with device draws
                self._draw_slot(j, s)
            if self.fused:
                self._fused_inner(j, lr, self.sse_log[self.n_prox % self.n_log])
                self.n_prox += 1
                self.s += 1
                return
            b.grad_stoch_diff(self.z, self.w, self.mbs, j, out=self.z, alpha=-lr / self.mb, beta=1.0, c1=self.z,
                              gamma=-lr, c2=self.mu)
        else:
            ops.axpbypcz(1.0, self.z, -lr, self.mu, out=self.z)
        self.z = self._prox(self.z)
        self.s += 1                                             # eager steps keep the index on the host (no counter launch)

    def _fused_outer(self, lr, sse_out):
        """outer refresh (mu = grad_full(z), w = z) + inner iteration 0 in one kernel (pnp_csmri_svrg_outer_step)."""
        b, px = self.b, self.prox
        b.plan.svrg_outer_step(self.z, b.bits, b.yh_full, b.inv_m0, lr, self.w, self.mu, out=self.z, denoise=px.fused_denoise,
                               xrec=b.xrec, sse=sse_out if px.fused_denoise else None, **px.fused_args())
        px.after_fused(self.z, b.xrec, sse_out)

    def _fused_inner(self, j, lr, sse_out):
        """step + estimate_sigma + prox + error of inner iteration j in one kernel (TV), or in one kernel + the network."""
        b, px = self.b, self.prox
        if self.mbs.host[j] is not None:                        # host-drawn selector: pack it to bits (a 5 us launch)
            self._hostbits = b.plan.pack_mask(self.mbs.host[j], out=self._hostbits)
            bits = self._hostbits
        else:
            bits = self.mbs.selbits[j]
        b.plan.svrg_step(self.z, self.w, bits, alpha=-lr / self.mb, beta=1.0, c1=self.z, gamma=-lr, c2=self.mu, out=self.z,
                         denoise=px.fused_denoise, xrec=b.xrec, sse=sse_out if px.fused_denoise else None, **px.fused_args())
        px.after_fused(self.z, b.xrec, sse_out)

    # ---- hipGraph form: one OUTER iteration (full-gradient refresh + T2 inner iterations) = one graph launch
    def _outer_body(self):
        b = self.b
        fold = self.variant == 'svrg' and self.fused and self.fold_outer
        if not fold:
            b.grad_full(self.z, out=self.mu)
            self.w.copy_(self.z)
        lr = self.eta
        if self.variant == 'svrg':
            b.draw(self.mbs, self.mb, self.seed, 0, self.T2, step_dev=self.step_dev)
        for j in range(self.T2):
            if fold and j == 0:
                self._fused_outer(lr, self.sse_tmp)
            elif self.variant == 'svrg' and self.fused:
                self._fused_inner(j, lr, self.sse_tmp)
            else:
                if self.variant == 'svrg':
                    b.grad_stoch_diff(self.z, self.w, self.mbs, j, out=self.z, alpha=-lr / self.mb, beta=1.0, c1=self.z,
                                      gamma=-lr, c2=self.mu)
                else:
                    ops.axpbypcz(1.0, self.z, -lr, self.mu, out=self.z)
                out = self.prox(self.z, b.xrec, self.sse_tmp)
                if out is not self.z:
                    raise ValueError('graph capture needs an in-place prox')
            ops.log_append(self.sse_tmp, self.sse_log, self.step_dev)
            ops.counter_add(self.step_dev, 1)

    def graph_ok(self):
        """Whether one outer iteration of this engine can be captured: constant step size, a prox that works in place and keeps
        no host-side per-call state (TVProx with denoise_strength == 0, DnCNNProx; not NLMProx, which ping-pongs)."""
        return (self.lr_decay == 1.0 and getattr(self.prox, 'inplace', False)
                and getattr(self.prox, 'denoise_strength', 0.0) == 0.0)

    def capture(self):
        """Capture one outer iteration into a hipGraph (torch.cuda.CUDAGraph on ROCm).  Needs `graph_ok()`, a step count that
        is a multiple of T2 and device-side minibatch draws.  State is left untouched, also when the capture fails."""
        if not self.graph_ok():
            raise ValueError('this engine cannot be captured in a hipGraph (needs lr_decay == 1 and an in-place prox without '
                             'host-side per-call state: TVProx with denoise_strength == 0 or DnCNNProx); step it eagerly')
        if not (self.s % self.T2 == 0 and self.n_prox == self.s):
            raise ValueError('capture() needs a step count that is a multiple of T2')
        self._set_dev_step(self.s)
        keep = (self.z.clone(), self.w.clone(), self.mu.clone(), self.sse_log.clone(), self.step_dev.clone())
        t_keep = getattr(self.prox, 't', None)
        z_obj = self.z
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._outer_body()                              # warm-up outside capture (lazy module loads)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._outer_body()
            torch.cuda.synchronize()
        finally:
            self.z = z_obj
            for dst, src in zip((self.z, self.w, self.mu, self.sse_log, self.step_dev), keep):
                dst.copy_(src)
            if t_keep is not None:
                self.prox.t = t_keep
        self.graph = g
        return g

    def _set_dev_step(self, s):
        """The device-resident counter is brought up to date only when a graph is about to read it."""
        if self._dev_step != s:
            self.step_dev.fill_(s)
            self._dev_step = s

    def outer_kernel_ok(self):
        """Whether whole outer iterations can run as ONE launch each (pnp_csmri_svrg_outer_iteration): the one-kernel iteration
        with the prox inside it (TV, no host-side per-call state), the outer refresh folded in, device-drawn minibatches."""
        return (self.fused and self.fold_outer and self.variant == 'svrg' and getattr(self.prox, 'fused_denoise', False)
                and getattr(self.prox, 'denoise_strength', 0.0) == 0.0 and all(h is None for h in self.mbs.host))

    def run_outer(self, n_outer=1, one_launch=None):
        """n_outer outer iterations = n_outer * T2 inner iterations, from a step count that is a multiple of T2.
        one_launch (default: when `outer_kernel_ok()`): every outer iteration is ONE draw launch + ONE kernel in which the
        workgroup that owns a problem runs its T2 inner iterations back to back -- the same bits as stepping; otherwise
        replays of the captured hipGraph."""
        if one_launch is None:
            one_launch = self.outer_kernel_ok()
        if one_launch:
            if not self.outer_kernel_ok() or self.s % self.T2 != 0 or self.n_prox != self.s:
                raise ValueError('one launch per outer iteration needs the one-kernel iteration with the TV prox, the folded '
                                 'refresh, device-drawn minibatches and a step count that is a multiple of T2')
            b, px = self.b, self.prox
            for _ in range(n_outer):
                b.draw(self.mbs, self.mb, self.seed, self.s, self.T2)
                lr = self.eta * self.lr_decay ** (self.s // self.T2)
                b.plan.svrg_outer_iteration(self.z, self.w, self.mu, b.bits, b.yh_full, b.inv_m0, self.mbs.selbits, self.T2, lr, self.mb,
                                            b.xrec, self.sse_log, self.n_prox % self.n_log, px.sig, sigma_modifier=px.sigma_modifier)
                self.s += self.T2
                self.n_prox += self.T2
                px.t += self.T2
            return
        if self.graph is None:
            self.capture()
        self._set_dev_step(self.s)
        for _ in range(n_outer):
            self.graph.replay()
            self.s += self.T2
            self.n_prox += self.T2
            if hasattr(self.prox, 't'):
                self.prox.t += self.T2
        self._dev_step = self.s


class SarahEngine(_StochEngine):
    """pnp_sarah over a batch (algorithms/pnp_sarah.py:28-104) with the quirks of v1: the outer step
    `w_next = prox(w_prev - eta * grad_full(z))` is logged but never adopted by z, w_next stays fixed through the
    inner loop, and the outer step ignores lr_decay.  One log row per prox: outer rows at s % T2 == 0."""

    def __init__(self, batch, prox, eta, T2, mini_batch_size, lr_decay=1.0, n_log=4096, seed=0):
        super().__init__(batch, prox, eta, mini_batch_size, lr_decay, n_log, seed)
        self.T2 = T2
        self.w_prev = torch.empty_like(self.z)
        self.w_next = torch.empty_like(self.z)
        self.v_prev = torch.empty_like(self.z)
        self.v_next = torch.empty_like(self.z)

    def step(self, idx_s=None):
        b, s = self.b, self.s
        if s % self.T2 == 0:
            self.w_prev.copy_(self.z)
            b.grad_full(self.z, out=self.v_prev)
            ops.axpbypcz(1.0, self.w_prev, -self.eta, self.v_prev, out=self.w_next)
            self.w_next = self._prox(self.w_next)
        j = self._minibatch(idx_s, s)
        b.grad_stoch_diff(self.w_next, self.w_prev, self.mbs, j, out=self.v_next, alpha=1.0 / self.mb, beta=1.0, c1=self.v_prev)
        lr = self.eta * self.lr_decay ** (s // self.T2)
        ops.axpbypcz(1.0, self.z, -lr, self.v_next, out=self.z)
        self.z = self._prox(self.z)
        self.v_prev, self.v_next = self.v_next, self.v_prev
        self.w_prev.copy_(self.z)
        self.s += 1


class SagaEngine(_StochEngine):
    """pnp_saga over a batch (algorithms/pnp_saga.py:25-72): a device table [hist][B][H][W] of minibatch
    gradients (all rows start as the first one), its running sum, and ONE fused kernel per step that replaces a
    row, updates the sum, forms v = g - prev + sum/hist and applies the step (`pnp_saga_table_update`).
    The replaced row r of every step is drawn on the host (one value per step for the whole batch; pass `r=` to
    `step` to impose the reference's `np.random.choice(hist_size, 1)` stream)."""

    def __init__(self, batch, prox, eta, mini_batch_size, hist_size=50, lr_decay=1.0, n_log=4096, seed=0, idx0=None):
        super().__init__(batch, prox, eta, mini_batch_size, lr_decay, n_log, seed)
        self.hist = hist_size
        self.g = torch.empty_like(self.z)
        self._rng = np.random.default_rng(seed + 977)
        # pnp_saga.py:25-31: one minibatch gradient at Xinit fills the whole table
        j = self._minibatch(idx0, 0xFFFFFFFF)
        batch.grad_stoch(self.z, self.mbs, j, out=self.g, alpha=1.0 / self.mb)
        self.table = self.g.unsqueeze(0).repeat(hist_size, 1, 1, 1).contiguous()
        self.tsum = ops.axpbypcz(float(hist_size), self.g, out=torch.empty_like(self.g))
        self.r_prev = 0

    def reset(self):
        raise NotImplementedError('build a new SagaEngine (the table initialisation is part of the constructor)')

    def step(self, idx_s=None, r=None):
        """r: the table row this step replaces -- one value for the whole batch, or one per problem (legacy-seeded sweeps:
        every item follows its own np.random stream, and with masks of different sizes the streams drift apart)."""
        j = self._minibatch(idx_s, self.s)
        if r is None:
            r = int(self._rng.integers(self.hist))
        self.b.grad_stoch(self.z, self.mbs, j, out=self.g, alpha=1.0 / self.mb)
        lr = self.eta * self.lr_decay ** self.s
        if np.ndim(r) == 0 and np.ndim(self.r_prev) == 0:
            r = int(r)
            ops.saga_table_update(self.z, self.g, self.table[r], self.table[self.r_prev], self.tsum, lr, 1.0 / self.hist)
        else:
            rv = np.broadcast_to(np.asarray(r, np.int64), (self.b.B,))
            pv = np.broadcast_to(np.asarray(self.r_prev, np.int64), (self.b.B,))
            for b in range(self.b.B):
                ops.saga_table_update(self.z[b], self.g[b], self.table[int(rv[b]), b], self.table[int(pv[b]), b], self.tsum[b],
                                      lr, 1.0 / self.hist)
            r = rv.copy()
        self.r_prev = r
        self.z = self._prox(self.z)
        self.s += 1


def make_engine(batch, prox, eta, T2, mini_batch_size, lr_decay=1.0, variant='svrg', algorithm='svrg', hist_size=50, **kw):
    """Engine by algorithm name: 'svrg' (default), 'sgd', 'gd', 'sarah', 'saga'."""
    if algorithm == 'svrg':
        return SvrgEngine(batch, prox, eta, T2, mini_batch_size, lr_decay=lr_decay, variant=variant, **kw)
    if algorithm == 'sarah':
        return SarahEngine(batch, prox, eta, T2, mini_batch_size, lr_decay=lr_decay, **kw)
    if algorithm == 'sgd':
        return SgdEngine(batch, prox, eta, mini_batch_size, lr_decay=lr_decay, **kw)
    if algorithm == 'saga':
        return SagaEngine(batch, prox, eta, mini_batch_size, hist_size=hist_size, lr_decay=lr_decay, **kw)
    if algorithm == 'gd':
        kw.pop('seed', None)
        return GdEngine(batch, prox, eta, lr_decay=lr_decay, **kw)
    raise ValueError(f'unknown algorithm {algorithm!r}')
