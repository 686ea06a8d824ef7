"""Oracle (test infrastructure): inverse-problem restatements in float64 NumPy.

Follows reference problems/problem.py, problems/CSMRI.py, problems/DeblurSR.py and
problems/PR.py.  All random draws come from the *global legacy* `np.random` stream in
the reference's call order, so `np.random.seed(s)` before construction reproduces the
reference's mask / noise / minibatches exactly.
"""
import numpy as np
from . import denoise as _dn


def load_image(img_path, H, W, img=None):
    """reference problems/problem.py:16-24: PIL open -> resize((H, W)) -> min-max."""
    if img is None:
        if img_path is None:
            raise Exception('Need to pass in image path or image')
        from PIL import Image
        img = np.array(Image.open(img_path).resize((H, W)))
    tmp = np.asarray(img)
    return (tmp - np.min(tmp)) / (np.max(tmp) - np.min(tmp))


class Problem:
    """reference problems/problem.py:8-129 (display/debug helpers left out)."""

    def __init__(self, img_path, H, W, img=None):
        self.H, self.W, self.N = H, W, H * W
        self.M = self.N
        self.Xrec = load_image(img_path, H, W, img)
        self.X = self.Xrec.ravel()
        self.Xinit = np.empty_like(self.X)

    def PSNR(self, w):                                   # problem.py:33-35
        return _dn.psnr(self.Xrec, w)

    def set_snr_sigma(self):                             # problem.py:37-46
        if self.snr is not None and self.sigma is None:
            self.sigma = self.get_sigma_from_snr()
        elif self.sigma is not None and self.snr is None:
            self.snr = self.get_snr_from_sigma()
        elif self.snr is None and self.sigma is None:
            self.sigma, self.snr = 0, 10e9
        else:
            raise Exception('Please specify either sigma (sigma) or signal-to-noise ratio (snr).')

    def get_snr_from_sigma(self):                        # problem.py:48-56
        if self.sigma > 0:
            return 10 * np.log10(np.linalg.norm(self.Y0.ravel()) / self.sigma ** 2 / self.H / self.W)
        if self.sigma == 0:
            return 10e9
        raise Exception('Sigma cannot be negative.')

    def get_sigma_from_snr(self):                        # problem.py:58-61 (norm, not norm^2)
        return np.sqrt(np.linalg.norm(self.Y0.ravel()) / 10 ** (self.snr / 10) / self.H / self.W)

    def select_mb(self, size):                           # problem.py:110-117
        if size > self.M:
            print('MB size is too big: ', size, ' > ', self.M)
        batch = np.zeros(self.M)
        batch[np.random.choice(self.M, size, replace=False)] = 1
        return batch.astype(int)


class CSMRI(Problem):
    """reference problems/CSMRI.py:11-89."""

    def __init__(self, img_path=None, H=256, W=256, sample_prob=0.5, snr=None, sigma=None, img=None):
        super().__init__(img_path, H, W, img)
        self.pname = 'csmri'
        self.sample_prob, self.snr, self.sigma = sample_prob, snr, sigma
        self.mask = np.random.choice([0, 1], size=(H, W), p=[1 - sample_prob, sample_prob])   # :43-45
        # :47-59 dense DFT matrix product F X F^T; equals fft2 for H == W
        i, j = np.meshgrid(np.arange(H), np.arange(W))
        self.F = np.power(np.exp(-2 * np.pi * 1J / H), i * j)
        self.Y0 = self.forward_model(self.X)
        self.set_snr_sigma()
        noises = np.random.normal(0, self.sigma, self.Y0.shape)                               # :32
        self.Y = self.Y0 + self.mask * noises
        xi = np.absolute(np.fft.ifft2(self.Y)).ravel()                                        # :35-36
        self.Xinit = (xi - np.min(xi)) / (np.max(xi) - np.min(xi))
        self.lrH, self.lrW = H, W
        self.M = self.N
        self.M0 = np.count_nonzero(self.mask)

    def forward_model(self, w):
        return self.mask * self.F.dot(w.reshape(self.H, self.W)).dot(self.F.T)

    def f(self, w):
        return np.linalg.norm(self.Y - self.forward_model(w)) ** 2 / 2 / self.M

    def select_mb(self, size):                           # CSMRI.py:66-74
        if size > self.M:
            print('MB size is too big: ', size, ' > ', self.M)
        batch = np.zeros(self.M)
        batch[np.random.choice(np.flatnonzero(self.mask), size, replace=False)] = 1
        return batch.reshape(self.H, self.W).astype(int)

    def _grad(self, z, sel):
        res = np.fft.fft2(z.reshape(self.H, self.W)) * sel
        idx = np.nonzero(sel)
        res[idx] = res[idx] - self.Y[idx]
        return np.real(np.fft.ifft2(res)).ravel()

    def grad_full(self, z):                              # CSMRI.py:76-81
        return self._grad(z, self.mask) / self.M0

    def grad_stoch(self, z, mb):                         # CSMRI.py:83-89 (un-normalised)
        return self._grad(z, self.mask * mb)


# ---------------------------------------------------------------------------------
# pylops 1.14.0 `signalprocessing.Bilinear` restated from its published semantics.
# pylops is not installed in the build container => "parity unpinned"; pinned only by
# the adjoint dot-test in tests/.
# ---------------------------------------------------------------------------------
class Bilinear:
    def __init__(self, iava, dims):
        self.dims = dims
        ncols = dims[1]
        r0 = np.floor(iava[0]).astype(int)
        c0 = np.floor(iava[1]).astype(int)
        wr = iava[0] - r0
        wc = iava[1] - c0
        self.taps = [(r0 * ncols + c0, (1 - wr) * (1 - wc)), ((r0 + 1) * ncols + c0, wr * (1 - wc)),
                     (r0 * ncols + c0 + 1, (1 - wr) * wc), ((r0 + 1) * ncols + c0 + 1, wr * wc)]
        self.n_out = iava.shape[1]

    def matvec(self, x):
        x = x.ravel()
        return sum(wt * x[idx] for idx, wt in self.taps)

    def rmatvec(self, y):
        out = np.zeros(self.dims[0] * self.dims[1])
        for idx, wt in self.taps:
            np.add.at(out, idx, wt * y)
        return out


class Deblur(Problem):
    """reference problems/DeblurSR.py:16-147."""
    EPS = 1e-10

    def __init__(self, img_path=None, H=64, W=64, kernel_path=None, kernel=None,
                 scale_percent=50, snr=None, sigma=None, img=None):
        super().__init__(img_path, H, W, img)
        self.pname = 'deblur'
        self.scale_percent, self.snr, self.sigma = scale_percent, snr, sigma
        if kernel_path is None and kernel is None:
            raise Exception('Need to pass in kernel path or kernel as image')
        if kernel_path is not None:                                                   # :72-93
            from PIL import Image
            B = np.array(Image.open(kernel_path).resize((H, W)))
        elif isinstance(kernel, str) and kernel == 'Identity':
            B = np.zeros(self.N)
            B[0] = 1
        elif isinstance(kernel, str) and kernel == 'Minimal':
            B = np.zeros((H, W))
            B[0, 0] = 1
            B[H // 2, H // 2] = 1
            B[H // 2, H // 3] = 1
            B[H // 2, H // 4] = 1
            B /= 4
        else:
            B = kernel
        self.B = np.asarray(B).ravel() / self.N
        self.lrH = int(H * scale_percent / 100)
        self.lrW = int(W * scale_percent / 100)
        self.M = self.lrH * self.lrW
        if scale_percent == 100:                                                      # :95-108
            self.Bop = None
        else:
            ptsH = np.linspace(self.EPS, H - (1 + self.EPS), self.lrH)
            ptsW = np.linspace(self.EPS, W - (1 + self.EPS), self.lrW)
            meshW, meshH = np.meshgrid(ptsH, ptsW)
            self.Bop = Bilinear(np.vstack([meshH.ravel(), meshW.ravel()]), (H, W))
        self.Y0 = self.forward_model(self.X)
        self.set_snr_sigma()
        self.Y = self.Y0 + np.random.normal(0, self.sigma, self.Y0.shape)
        self.Xinit = np.random.uniform(0.0, 1.0, self.N)

    def _down(self, x):
        return x if self.Bop is None else self.Bop.matvec(x)

    def _up(self, y):
        return y if self.Bop is None else self.Bop.rmatvec(y)

    def fft_blur(self, a, b):                                                         # :119-120
        return np.real(np.fft.ifft(np.fft.fft(a.ravel()) * np.fft.fft(b.ravel()))) * np.sqrt(self.N)

    def forward_model(self, w):
        return self._down(self.fft_blur(w, self.B))

    def f(self, w):
        return np.linalg.norm(self.Y - self.forward_model(w)) ** 2 / 2 / self.M

    def grad_full(self, z):                                                           # :126-132
        res = self._down(self.fft_blur(z.ravel(), self.B)) - self.Y
        return self.fft_blur(self._up(res), np.roll(np.flip(self.B), 1)) / self.M

    def grad_stoch(self, z, mb):                                                      # :135-147
        idx = np.nonzero(mb.ravel())
        res = np.zeros(self.M)
        down = self._down(self.fft_blur(z.ravel(), self.B))
        res[idx] = down[idx] - self.Y[idx]
        return self.fft_blur(self._up(res), np.roll(np.flip(self.B), 1))


class PhaseRetrieval(Problem):
    """reference problems/PR.py:12-87."""

    def __init__(self, img_path=None, H=256, W=256, num_meas=-1, snr=None, sigma=None, img=None):
        super().__init__(img_path, H, W, img)
        self.pname = 'pr'
        self.M, self.snr, self.sigma = num_meas, snr, sigma
        self.A = np.random.randn(self.M, self.N)
        self.Y0 = self.forward_model(self.X).ravel()
        self.set_snr_sigma()
        self.Y = self.Y0 + np.random.normal(0, self.sigma, self.Y0.shape)
        self.spec_init()
        self.Xinit = (self.Xinit - self.Xinit.min()) / (self.Xinit.max() - self.Xinit.min())

    def spec_init(self):                                                              # :50-63
        nrm = np.linalg.norm(self.X)
        D = self.A.T.dot(self.A * self.Y[:, None]) / self.M
        m, mold = 1, 2
        y_final, y_old = 2 * np.ones(self.N), np.ones(self.N)
        tol = 1e-5
        while abs(m - mold) > tol and np.linalg.norm(y_final - y_old) > tol:
            mold, y_old = m, y_final
            y_final = D.dot(y_final)
            m = np.max(y_final)
            y_final = y_final / m
        self.Xinit = np.sqrt(m) * y_final / np.linalg.norm(y_final) * nrm

    def forward_model(self, w):
        return np.absolute(self.A.dot(w))

    def f(self, w):
        return np.linalg.norm(self.Y - self.forward_model(w)) ** 2 / 2 / self.M

    def grad_full(self, z):                                                           # :75-79
        t = self.A.dot(z.ravel()).ravel()
        wt = (np.absolute(t) - self.Y.ravel()) / np.absolute(t)
        return self.A.T.dot(wt * t).ravel() / self.M

    def grad_stoch(self, z, mb):                                                      # :81-87
        idx = np.nonzero(mb)
        Ag = self.A[idx]
        t = Ag.dot(z.ravel()).ravel()
        wt = (np.absolute(t) - self.Y[idx]) / np.absolute(t)
        return Ag.T.dot(wt * t).ravel()
