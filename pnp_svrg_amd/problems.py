"""Host-side mirror of the reference's `problems/*` interface, backed by the HIP kernels.

Same names, constructor arguments, attributes, return conventions and error behaviour as
reference problems/problem.py, problems/CSMRI.py; the data-fidelity gradients
run on the MI355X through the C ABI.  Problem *construction* (image load, random mask /
noise from the global legacy `np.random` stream, forward model) is one-off setup and stays
in NumPy float64 so that a seed reproduces the reference's data bit for bit.

Extensions (keyword-only, all optional): `img=` (pixel array instead of a path),
`dtype=` (torch.float32 production / torch.float64 parity), `device=` (default: the current HIP device).
Gradient methods accept either a NumPy vector (returns NumPy float64, like the reference)
or a device tensor (returns a device tensor: the loops use this form and never leave HBM).
"""
import numpy as np
import torch

from . import legacy_rng, ops

_DEFAULT_DTYPE = torch.float32


def set_default_dtype(dtype):
    """torch.float32 (production) or torch.float64 (parity/debug) for newly built problems."""
    global _DEFAULT_DTYPE
    assert dtype in (torch.float32, torch.float64)
    _DEFAULT_DTYPE = dtype


def get_default_dtype():
    return _DEFAULT_DTYPE


class Problem():
    """reference problems/problem.py:8-129."""

    def __init__(self, img_path, H, W, *, img=None, dtype=None, device=None, upload=True):
        self.H = H
        self.W = W
        self.N = H * W
        self.M = self.N
        if img is not None:
            tmp = np.asarray(img)
        elif img_path is not None:
            from PIL import Image
            tmp = np.array(Image.open(img_path).resize((H, W)))
        else:
            raise Exception('Need to pass in image path or image')
        tmp = (tmp - np.min(tmp)) / (np.max(tmp) - np.min(tmp))
        self.Xrec = tmp
        self.X = tmp.ravel()
        self.Xinit = np.empty_like(self.X)
        # device side (upload=False: host-side construction only -- the batched engines upload whole batches)
        self.dtype = dtype if dtype is not None else _DEFAULT_DTYPE
        if device is None:
            # the CURRENT device: one process per GPU under torchrun has set it to its LOCAL_RANK, so a problem built on
            # rank r lives on GPU r (a fixed 'cuda:0' default made every rank but 0 raise below)
            device = torch.device('cuda', torch.cuda.current_device()) if (upload and torch.cuda.is_available()) else torch.device('cuda')
        self.device = torch.device(device)
        self._upload_enabled = upload
        if upload:
            ops.require_gpu()
            if self.device.type == 'cuda' and self.device.index not in (None, torch.cuda.current_device()):
                # plans allocate on, and kernels launch on, the CURRENT HIP device
                raise Exception(f'device {self.device} is not the current device (cuda:{torch.cuda.current_device()}); '
                                'call torch.cuda.set_device first')
            self._xrec_d = self.to_device(self.Xrec).reshape(1, H, W)

    # ---- host <-> device helpers (extensions)
    def to_device(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=self.dtype).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device=self.device, dtype=self.dtype)

    @staticmethod
    def _is_dev(a):
        return isinstance(a, torch.Tensor)

    def _ret(self, t, like):
        return t if self._is_dev(like) else t.reshape(-1).double().cpu().numpy()

    def get_item(self, key):
        return self.__dict__[key]

    def sse_device(self, w_dev):
        """sum (Xrec - w)^2 on the device; float64 scalar tensor [1]."""
        return ops.sse(w_dev.reshape(1, self.H, self.W), self._xrec_d)

    @staticmethod
    def psnr_from_sse(sse, n):
        """reference problems/problem.py:33-35: 10 log10(1/mse), rounded to 2 decimals."""
        with np.errstate(divide='ignore'):
            return np.around(10 * np.log10(1.0 / (np.float64(sse) / n)), decimals=2)

    def PSNR(self, w):
        w_dev = w if self._is_dev(w) else self.to_device(w)
        return self.psnr_from_sse(self.sse_device(w_dev).item(), self.N)

    def set_snr_sigma(self):
        if self.snr is not None and self.sigma is None:
            self.sigma = self.get_sigma_from_snr()
        elif self.sigma is not None and self.snr is None:
            self.snr = self.get_snr_from_sigma()
        elif self.snr is None and self.sigma is None:
            self.sigma = 0
            self.snr = 10e9
        else:
            raise Exception('Please specify either sigma (sigma) or signal-to-noise ratio (snr).')

    def get_snr_from_sigma(self):
        if self.sigma > 0:
            return 10 * np.log10(np.linalg.norm(self.Y0.ravel()) / self.sigma ** 2 / self.H / self.W)
        elif self.sigma == 0:
            return 10e9
        raise Exception('Sigma cannot be negative.')

    def get_sigma_from_snr(self):
        # the reference divides the NORM (not its square) by the linear SNR; kept as is
        return np.sqrt(np.linalg.norm(self.Y0.ravel()) / 10 ** (self.snr / 10) / self.H / self.W)

    def display(self, color_map='gray', show_measurements=False, save_results=False, save_dir='figures/', show_figs=False):
        """reference problems/problem.py:64-108 (host-side matplotlib convenience, not on the hot path): shows /
        saves the original image, the initialisation and optionally the measurements, and -- what callers such
        as Utilities.display_results rely on -- sets `color_map` and `prob_dir`."""
        self.color_map = color_map
        import matplotlib.pyplot as plt
        base = None
        if save_results:
            from datetime import datetime
            import os
            base = save_dir + self.pname + '/' + datetime.now().strftime('%y-%m-%d-%H-%M') + "/"
            self.prob_dir = base
            os.makedirs(base, exist_ok=True)
        panels = [('Original Image', self.Xrec, 'original.eps'),
                  ('Initialization', self.Xinit.reshape(self.H, self.W), 'initialization.eps')]
        if show_measurements:
            panels.append(('Measurements', np.real(self.Y).reshape(self.lrH, self.lrW), 'measurements.eps'))
        for title, img, fname in panels:
            fig = plt.figure(figsize=(6, 6))
            plt.imshow(img, cmap=color_map, vmin=0, vmax=1)
            plt.title(title)
            plt.xticks([])
            plt.yticks([])
            if save_results:
                fig.savefig(base + fname, transparent=True, bbox_inches='tight', pad_inches=0)
            if show_figs:
                plt.show()
            plt.close(fig)

    def select_mb(self, size):
        if size > self.M:
            print('MB size is too big: ', size, ' > ', self.M)
        batch_locs = legacy_rng.choice(self.M, size)              # = np.random.choice(self.M, size, replace=False)
        batch = np.zeros(self.M, dtype=int)                       # (the reference fills a float vector and casts: same array)
        batch[batch_locs] = 1
        return batch

    def f(self, z):
        raise NotImplementedError('Need to implement f() method')

    def grad_full(self, z):
        raise NotImplementedError('Need to implement full_grad() method')

    def grad_stoch(self, z, mb_indices):
        raise NotImplementedError('Need to implement stoch_grad() method')


class CSMRI(Problem):
    """reference problems/CSMRI.py:11-89 with the gradients on the MI355X."""

    def __init__(self, img_path=None, H=256, W=256, sample_prob=0.5, snr=None, sigma=None, **ext):
        super().__init__(img_path, H, W, **ext)
        self.pname = 'csmri'
        self.sample_prob = sample_prob
        self.snr = snr
        self.sigma = sigma

        self._generate_mask()
        self.Y0 = self.forward_model(self.X)
        self.set_snr_sigma()
        noises = np.random.normal(0, self.sigma, self.Y0.shape)
        self.Y = self.Y0 + np.multiply(self.mask, noises)
        self.SNR = self.get_snr_from_sigma
        self.Xinit = np.absolute(np.fft.ifft2(self.Y)).ravel()
        self.Xinit = (self.Xinit - np.min(self.Xinit)) / (np.max(self.Xinit) - np.min(self.Xinit))
        self.lrH, self.lrW = self.H, self.W
        self.M = self.N
        self.M0 = np.count_nonzero(self.mask)
        if self._upload_enabled:
            self._upload()

    def _generate_mask(self):
        self.mask = np.random.choice([0, 1], size=(self.H, self.W), p=[1 - self.sample_prob, self.sample_prob])

    def forward_model(self, w):
        # reference CSMRI.py:53-59 multiplies by a dense DFT matrix (== fft2 to 3e-11 for H == W)
        return np.multiply(self.mask, np.fft.fft2(w.reshape(self.H, self.W)))

    def f(self, w):
        w = w.double().cpu().numpy() if self._is_dev(w) else w
        return np.linalg.norm(self.Y - self.forward_model(w)) ** 2 / 2 / self.M

    # ---- device state
    def _upload(self):
        H, W = self.H, self.W
        cdt = torch.complex64 if self.dtype == torch.float32 else torch.complex128
        self.plan = ops.CsmriPlan(H, W, 1, self.dtype)
        self._mask_idx = np.flatnonzero(self.mask).astype(np.int32)
        self._YT = torch.from_numpy(np.ascontiguousarray(self.Y.T)).to(self.device, cdt).reshape(1, W, H).contiguous()
        idx = torch.from_numpy(self._mask_idx).to(self.device).reshape(1, -1)
        self._maskT = self.plan.sel_from_indices(idx)
        self._yh_full = self.plan.pack_y(self._YT, self._maskT)
        self._selT = torch.empty_like(self._maskT)

    def select_mb(self, size):
        if size > self.M:
            print('MB size is too big: ', size, ' > ', self.M)
        mask_locs = self._locs()
        batch_locs = legacy_rng.choice(mask_locs, size)           # = np.random.choice(mask_locs, size, replace=False)
        mb = np.zeros((self.H, self.W), dtype=int)                # (the reference fills a float vector and casts: same array)
        mb.reshape(-1)[batch_locs] = 1
        self._last_mb = (mb, batch_locs)                          # grad_stoch(z, mb) with this very array skips its flatnonzero
        return mb

    def _locs(self):
        """np.flatnonzero(self.mask) (CSMRI.py:71), computed once.  ASSUMPTION: the mask is frozen after construction -- the
        reference recomputes it on every select_mb; code that edits `problem.mask` afterwards must delete `problem._mask_locs`."""
        locs = self.__dict__.get('_mask_locs')
        if locs is None:
            locs = self._mask_locs = np.asarray(np.flatnonzero(self.mask))
        return locs

    def _select_mb_locs(self, size):
        """The draw of select_mb without building the H x W indicator: the same np.random.choice call on the same
        (cached) array of sampled locations, so the legacy stream advances identically."""
        if size > self.M:
            print('MB size is too big: ', size, ' > ', self.M)
        return legacy_rng.choice(self._locs(), size)

    def _selector(self, mb):
        """mask o mb (CSMRI.py:84) -> transposed device selector."""
        last = self.__dict__.get('_last_mb')
        if (last is not None and last[0] is mb and np.count_nonzero(mb) == last[1].shape[0]
                and mb.reshape(-1)[last[1]].all()):             # (same count + every drawn location still set = the same set)
            sel = last[1].astype(np.int32)                        # select_mb's own locations: all inside the mask
        else:
            sel = np.flatnonzero(np.multiply(self.mask, np.asarray(mb).reshape(self.H, self.W))).astype(np.int32)
        return self.plan.sel_from_indices(self._upload_idx(sel), out=self._selT)

    def _upload_idx(self, sel):
        """int32 index list -> device [1, n] through a pinned staging buffer (an asynchronous 4-KB copy instead of a blocking
        pageable one: 6 against 22 us); the event keeps the next call from overwriting a copy still in flight."""
        n = sel.shape[0]
        st = self.__dict__.get('_idx_stage')
        if st is None or st[0].shape[0] != n:
            st = self._idx_stage = (torch.empty(n, dtype=torch.int32, pin_memory=True),
                                    torch.empty((1, n), dtype=torch.int32, device=self.device), torch.cuda.Event())
        else:
            st[2].synchronize()
        st[0].numpy()[:] = sel
        st[1].copy_(st[0].reshape(1, n), non_blocking=True)
        st[2].record()
        return st[1]

    def grad_full(self, z):
        zd = (z if self._is_dev(z) else self.to_device(z)).reshape(1, self.H, self.W)
        g = self.plan.grad(zd, self._maskT, yh=self._yh_full, alpha=1.0 / self.M0)
        return self._ret(g.reshape(-1), z)

    def grad_stoch(self, z, mb, *, scale=1.0):
        zd = (z if self._is_dev(z) else self.to_device(z)).reshape(1, self.H, self.W)
        selT = self._selector(mb)
        g = self.plan.grad(zd, selT, YT=self._YT, alpha=scale)      # the selector's data term is formed in the column pass
        return self._ret(g.reshape(-1), z)

    def grad_stoch_diff(self, z, w, mb, alpha=1.0, beta=0.0, c1=None, gamma=0.0, c2=None, out=None):
        """alpha*(grad_stoch(z,mb) - grad_stoch(w,mb)) + beta*c1 + gamma*c2 in ONE FFT pair
        (the Y terms cancel).  Device tensors only."""
        shp = (1, self.H, self.W)
        selT = self._selector(mb)
        return self.plan.grad(z.reshape(shp), selT, b=w.reshape(shp), alpha=alpha, beta=beta, c1=c1,
                              gamma=gamma, c2=c2, out=out)


def _bilinear_taps(iava, dims):
    """pylops 1.14 `signalprocessing.Bilinear(iava, dims)` restated from its published semantics
    (floor index + fractional weights, 4 taps; adjoint = transposed taps).  pylops itself is not
    available in the build container: parity of this operator is UNPINNED (adjoint dot-test only)."""
    ncols = dims[1]
    r0 = np.floor(iava[0]).astype(np.int64)
    c0 = np.floor(iava[1]).astype(np.int64)
    wr, wc = iava[0] - r0, iava[1] - c0
    idx = np.stack([r0 * ncols + c0, (r0 + 1) * ncols + c0, r0 * ncols + c0 + 1, (r0 + 1) * ncols + c0 + 1], axis=1)
    wts = np.stack([(1 - wr) * (1 - wc), wr * (1 - wc), (1 - wr) * wc, wr * wc], axis=1)
    M, N = idx.shape[0], dims[0] * dims[1]
    # CSR of the adjoint: for every image pixel the (measurement, weight) pairs that touch it
    flat_t = idx.ravel()
    order = np.argsort(flat_t, kind='stable')
    a_col = (np.arange(M * 4) // 4)[order].astype(np.int32)
    a_val = wts.ravel()[order]
    a_rowptr = np.zeros(N + 1, np.int32)
    np.add.at(a_rowptr, flat_t + 1, 1)
    a_rowptr = np.cumsum(a_rowptr).astype(np.int32)
    return idx.astype(np.int32), wts, a_rowptr, a_col, a_val


class Deblur(Problem):
    """reference problems/DeblurSR.py:16-147 with forward model and gradients on the MI355X."""
    eps = 1e-10

    def __init__(self, img_path=None, H=64, W=64, kernel_path=None, kernel=None, scale_percent=50, snr=None,
                 sigma=None, **ext):
        super().__init__(img_path, H, W, **ext)
        self.pname = 'deblur'
        self.scale_percent = scale_percent
        self.snr = snr
        self.sigma = sigma
        self.kernel_path = kernel_path
        self.kernel = kernel
        if kernel_path is None and kernel is None:
            raise Exception('Need to pass in kernel path or kernel as image')
        self._load_kernel()
        self.lrH = int(self.H * scale_percent / 100)
        self.lrW = int(self.W * scale_percent / 100)
        self.M = self.lrH * self.lrW
        self._generate_bop()
        self.Y0 = self.forward_model(self.X)
        self.set_snr_sigma()
        noises = np.random.normal(0, self.sigma, self.Y0.shape)
        self.Y = self.Y0 + noises
        self.Xinit = np.random.uniform(0.0, 1.0, self.N)
        self._Y_d = self.to_device(self.Y).reshape(1, self.M)

    def _load_kernel(self):
        if self.kernel_path is not None:
            from PIL import Image
            self.B = np.array(Image.open(self.kernel_path).resize((self.H, self.W)))
        elif isinstance(self.kernel, str) and self.kernel == "Identity":
            self.B = np.zeros(self.N)
            self.B[0] = 1
        elif isinstance(self.kernel, str) and self.kernel == "Minimal":
            self.B = np.zeros((self.H, self.W))
            self.B[0, 0] = 1
            self.B[self.H // 2, self.H // 2] = 1
            self.B[self.H // 2, self.H // 3] = 1
            self.B[self.H // 2, self.H // 4] = 1
            self.B /= 4
        elif self.kernel is not None:
            self.B = self.kernel
        else:
            raise Exception('Need to pass in blur kernel path or kernel')
        self.B = np.asarray(self.B).ravel() / self.N

    def _generate_bop(self):
        taps = None
        if self.scale_percent != 100:
            ptsH = np.linspace(self.eps, self.H - (1 + self.eps), self.lrH)
            ptsW = np.linspace(self.eps, self.W - (1 + self.eps), self.lrW)
            meshW, meshH = np.meshgrid(ptsH, ptsW)                    # (sic) as DeblurSR.py:102
            iava = np.vstack([meshH.ravel(), meshW.ravel()])
            taps = _bilinear_taps(iava, (self.H, self.W))
        self.Bop = taps                                               # forward taps + CSR adjoint (None = Identity)
        self.plan = ops.DeblurPlan(self.H, self.W, 1, self.dtype, self.B, bilinear=taps)

    def forward_model(self, w):
        wd = (w if self._is_dev(w) else self.to_device(w)).reshape(1, self.N)
        y = self.plan.forward(wd)
        return y if self._is_dev(w) else y.double().cpu().numpy()

    def f(self, w):
        w = w.double().cpu().numpy() if self._is_dev(w) else w
        return np.linalg.norm(self.Y - self.forward_model(w)) ** 2 / 2 / self.M

    def grad_full(self, z):
        zd = (z if self._is_dev(z) else self.to_device(z)).reshape(1, self.N)
        g = self.plan.grad(zd, self._Y_d, scale=1.0 / self.M)
        return self._ret(g.reshape(-1), z)

    def grad_stoch(self, z, mb, *, scale=1.0):
        zd = (z if self._is_dev(z) else self.to_device(z)).reshape(1, self.N)
        sel = torch.from_numpy(np.ascontiguousarray(np.asarray(mb).ravel() != 0).astype(np.uint8)).to(self.device).reshape(1, self.M)
        g = self.plan.grad(zd, self._Y_d, sel=sel, scale=scale)
        return self._ret(g.reshape(-1), z)


class PhaseRetrieval(Problem):
    """reference problems/PR.py:12-87 with the amplitude-flow gradients on the MI355X.  Setup
    (Gaussian A from the legacy np.random stream, PR.py:26-35) stays in NumPy float64; the spectral
    initialisation (PR.py:50-63) is a device-side power iteration that never forms A^T diag(Y) A."""

    def __init__(self, img_path=None, H=256, W=256, num_meas=-1, snr=None, sigma=None, **ext):
        super().__init__(img_path, H, W, **ext)
        self.pname = 'pr'
        self.M = num_meas
        self.snr = snr
        self.sigma = sigma
        self.A = np.random.randn(self.M, self.N)
        self.Y0 = np.absolute(self.A.dot(self.X)).ravel()
        self.set_snr_sigma()
        noises = np.random.normal(0, self.sigma, self.Y0.shape)
        self.Y = self.Y0 + noises
        self.SNR = self.get_snr_from_sigma
        self.spec_init()
        self.Xinit = (self.Xinit - self.Xinit.min()) / (self.Xinit.max() - self.Xinit.min())
        self._A_d = self._A64_d if self.dtype == torch.float64 else self._A64_d.to(self.dtype)
        del self._A64_d
        self._Y_d = self.to_device(self.Y)
        self._ws = None

    def spec_init(self):
        """reference PR.py:50-63: leading eigenvector of D = A^T diag(Y) A / M by normalised power iteration, scaled
        to the image norm.  On the device D is never formed (N x N; 2 GiB and ~4 TFLOP at the notebooks' 128 x 128):
        every step is one `pnp_pr_spectral_apply` (A streamed twice), always float64 like the reference; the
        stopping rule (both |m - m_old| and the iterate change above 1e-5) is evaluated on the host each step."""
        ops.require_gpu()
        A64 = torch.from_numpy(self.A).to(self.device)
        Y64 = torch.from_numpy(np.ascontiguousarray(self.Y, dtype=np.float64)).to(self.device)
        from . import _native as N
        ws = torch.empty(N.lib().pnp_pr_workspace_elems(self.M, self.N), dtype=torch.float64, device=self.device)
        v = torch.full((self.N,), 2.0, dtype=torch.float64, device=self.device)
        prev = torch.ones_like(v)
        lead, lead_old, tol = 1, 2, 1e-5
        while abs(lead - lead_old) > tol and float(torch.linalg.vector_norm(v - prev)) > tol:
            lead_old, prev = lead, v
            v = ops.pr_spectral_apply(A64, prev, Y64, scale=1.0 / self.M, workspace=ws)
            lead = float(v.max())
            v = v / lead
        v = v.cpu().numpy()
        self.Xinit = np.sqrt(lead) * v / np.linalg.norm(v) * np.linalg.norm(self.X)
        self._A64_d = A64                       # reused below when the problem itself runs in float64

    def forward_model(self, w):
        w = w.double().cpu().numpy() if self._is_dev(w) else w
        return np.absolute(self.A.dot(w))

    def f(self, w):
        return np.linalg.norm(self.Y - self.forward_model(w)) ** 2 / 2 / self.M

    def _grad(self, z, rows, scale):
        zd = (z if self._is_dev(z) else self.to_device(z)).reshape(-1)
        if self._ws is None:
            from . import _native as N
            self._ws = torch.empty(N.lib().pnp_pr_workspace_elems(self.M, self.N), dtype=self.dtype, device=self.device)
        g = ops.pr_grad(self._A_d, zd, self._Y_d, rows=rows, scale=scale, workspace=self._ws)
        return self._ret(g, z)

    def grad_full(self, z):
        return self._grad(z, None, 1.0 / self.M)

    def grad_stoch(self, z, mb, *, scale=1.0):
        rows = torch.from_numpy(np.flatnonzero(np.asarray(mb)).astype(np.int32)).to(self.device)
        return self._grad(z, rows, scale)
