"""Oracle (test infrastructure): prox / noise-estimate / PSNR restatements in float64 NumPy.

Third-party semantics restated here (pinned bit-for-bit or to <=1e-12 by
tests/test_oracle_golden.py against outputs of the real libraries):

* skimage 0.18 `estimate_sigma(img2d, multichannel=True, average_sigmas=True)`
  as called at reference algorithms/pnp_svrg.py:71 (pnp_gd.py:49, pnp_sgd.py:50,
  pnp_saga.py:64, pnp_sarah.py:46,90).
* skimage 0.18 `denoise_wavelet(img2d, method='BayesShrink', sigma=s,
  multichannel=True)` as called by reference denoisers/TV.py:21-26.
* skimage 0.18 `denoise_nl_means(..., fast_mode=False)` as called by reference
  denoisers/NLM.py:22-27 (Cython `_nl_means_denoising_2d`, Schraudolph fast_exp).
* skimage `peak_signal_noise_ratio` + `np.around(.,2)`, reference
  problems/problem.py:33-35.
* the RealSN_DnCNN wrapper arithmetic, reference denoisers/RealSN_DnCNN.py:16-42,
  around the 17-layer net of denoisers/DeepDenoisers/model/models.py:5-22.
"""
import numpy as np

# PyWavelets 1.1.1 filter banks (pywt.Wavelet('db2').dec_hi, Wavelet('db1'))
DB2_DEC_HI = (-0.48296291314453416, 0.8365163037378079,
              -0.2241438680420134, -0.12940952255126037)
HAAR = 0.7071067811865476
MAD_DENOM = 0.6744897501960817          # scipy.stats.norm.ppf(0.75)


# --------------------------------------------------------------------------
# estimate_sigma
# --------------------------------------------------------------------------
def db2_detail_cols(img):
    """Single-level db2 detail coefficients of every column (axis 0), pywt mode
    'symmetric'.  (H, W) -> ((H+3)//2, W).  cD[i] = sum_j hi[j]*x[2i+1-j]."""
    x = np.asarray(img, dtype=np.float64)
    H = x.shape[0]
    n = (H + 3) // 2
    xe = np.pad(x, ((3, 3), (0, 0)), mode='symmetric')
    i2 = 2 * np.arange(n) + 3                       # index of x[2i] inside xe
    h0, h1, h2, h3 = DB2_DEC_HI
    return ((h0 * xe[i2 + 1] + h1 * xe[i2]) + h2 * xe[i2 - 1]) + h3 * xe[i2 - 2]


def sigma_cols(img):
    """Per-column MAD sigma: median(|d[d != 0]|) / 0.6745 (skimage _sigma_est_dwt)."""
    d = np.abs(db2_detail_cols(img))
    d = np.where(d == 0.0, np.nan, d)
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            med = np.nanmedian(d, axis=0)
    return med / MAD_DENOM


def estimate_sigma(img):
    """skimage estimate_sigma(img, multichannel=True, average_sigmas=True) on a 2-D
    image: the last axis is taken as channels, so it is the mean over the W columns
    of a 1-D estimate."""
    return float(np.mean(sigma_cols(img)))


# --------------------------------------------------------------------------
# "TV" denoiser = per-column 1-D Haar BayesShrink
# --------------------------------------------------------------------------
def haar_levels(H):
    """max(pywt.dwtn_max_level((H,), 'db1') - 3, 1)."""
    return max(int(np.floor(np.log2(H))) - 3, 1)


def haar_bayes_cols(img, sigma):
    """denoise_wavelet(img, method='BayesShrink', sigma=sigma, multichannel=True,
    mode='soft', wavelet='db1') on a float 2-D image: every column independently."""
    a = np.asarray(img, dtype=np.float64)
    L = haar_levels(a.shape[0])
    var = float(sigma) ** 2
    eps = np.finfo(np.float64).eps
    details = []
    for _ in range(L):
        ev, od = a[0::2], a[1::2]
        details.append(-HAAR * od + HAAR * ev)
        a = HAAR * od + HAAR * ev
    with np.errstate(all='ignore'):
        for d in reversed(details):
            dvar = np.mean(d * d, axis=0)
            thr = var / np.sqrt(np.maximum(dvar - var, eps))
            shrink = 1.0 - thr[None, :] / np.abs(d)
            np.clip(shrink, 0.0, None, out=shrink)
            d = d * shrink
            up = np.empty((2 * a.shape[0], a.shape[1]))
            up[0::2] = HAAR * a + HAAR * d
            up[1::2] = HAAR * a - HAAR * d
            a = up
    return a


class TVDenoiser:
    """reference denoisers/TV.py:9-26."""

    def __init__(self, multi=True, rescale_sigma=True, decay=1,
                 denoise_strength=0, sigma_modifier=1):
        self.t = 0
        self.decay = decay
        self.denoise_strength = denoise_strength
        self.sigma_modifier = sigma_modifier

    def denoise(self, noisy, sigma_est=0):
        self.t += 1
        if sigma_est > 0:
            s = sigma_est * self.sigma_modifier
        else:
            s = self.denoise_strength * self.decay ** self.t
        return haar_bayes_cols(noisy, s)


# --------------------------------------------------------------------------
# NLM (slow mode, 2-D, one channel) with Schraudolph fast_exp
# --------------------------------------------------------------------------
def fast_exp(y):
    """skimage fast_exp.h: high int32 word = (int32)(2^20/ln2 * y) + (1072693248-60801),
    low word 0, reinterpret as double.  Vectorised; C truncation toward zero."""
    y = np.asarray(y, dtype=np.float64)
    prod = np.trunc(1512775.3951951856938 * y)
    with np.errstate(invalid='ignore'):
        # (int) of an out-of-range double is INT_MIN on x86 (cvttsd2si), which is what the compiled kernel does
        prod = np.where(np.abs(prod) < 2.0 ** 31, prod, -2.0 ** 31)
    hi = prod.astype(np.int64) + 1072632447
    hi = ((hi + 2 ** 31) % 2 ** 32 - 2 ** 31).astype(np.int64)     # int32 wrap of the addition
    bits = (hi.astype(np.int64) << 32).astype(np.int64)
    return bits.view(np.float64)


def nl_means_2d(img, h, sigma, patch_size=4, patch_distance=5):
    """_nl_means_denoising_2d(image[...,None], s, d, h, var=sigma^2)."""
    x = np.asarray(img, dtype=np.float64)
    s = patch_size + 1 if patch_size % 2 == 0 else patch_size
    d = patch_distance
    off = s // 2
    H, W = x.shape
    pad = np.pad(x, off, mode='reflect')
    A = (s - 1.0) / 4.0
    g = np.arange(-off, off + 1)
    gr, gc = np.meshgrid(g, g, indexing='ij')
    w = np.exp(-(gr * gr + gc * gc) / (2 * A * A))
    w = w * (1.0 / (1 * np.sum(w) * h * h))
    var = 2.0 * sigma * sigma
    rows = np.arange(H)[:, None]
    cols = np.arange(W)[None, :]
    wsum = np.zeros((H, W))
    acc = np.zeros((H, W))
    for di in range(-d, d + 1):
        for dj in range(-d, d + 1):
            valid = ((rows + di >= 0) & (rows + di < H) &
                     (cols + dj >= 0) & (cols + dj < W))
            # clipped neighbour coordinates (invalid ones are masked out below)
            ri = np.clip(rows + di, 0, H - 1)
            cj = np.clip(cols + dj, 0, W - 1)
            dist = np.zeros((H, W))
            dead = np.zeros((H, W), dtype=bool)
            for pi in range(s):
                dead |= dist > 5.0
                for pj in range(s):
                    diff = pad[rows + pi, cols + pj] - pad[ri + pi, cj + pj]
                    dist = dist + w[pi, pj] * (diff * diff - var)
            weight = fast_exp(-np.maximum(0.0, dist))
            weight = np.where(dead, 0.0, weight)
            weight = np.where(valid, weight, 0.0)
            wsum = wsum + weight
            acc = acc + weight * pad[ri + off, cj + off]
    return acc / wsum


class NLMDenoiser:
    """reference denoisers/NLM.py:9-27.  The reference reads `self.sigma`, which its
    constructor never sets; callers must assign it.  Same here."""

    def __init__(self, decay=1, denoise_strength=0, patch_size=4, patch_distance=5,
                 sigma_modifier=1, fast_mode=False, multichannel=True):
        self.t = 0
        self.decay = decay
        self.denoise_strength = denoise_strength
        self.sigma_modifier = sigma_modifier
        self.patch_size = patch_size
        self.patch_distance = patch_distance

    def denoise(self, noisy, sigma_est=0):
        self.t += 1
        if self.sigma > 0:                       # AttributeError unless set: as reference
            hs = sigma_est * self.sigma_modifier
            return nl_means_2d(noisy, hs, hs, self.patch_size, self.patch_distance)
        hh = self.denoise_strength * self.decay ** self.t
        return nl_means_2d(noisy, hh, 0.0, self.patch_size, self.patch_distance)


# --------------------------------------------------------------------------
# PSNR
# --------------------------------------------------------------------------
def psnr_raw(xrec, w):
    err = np.mean((np.asarray(xrec, np.float64) - np.asarray(w, np.float64).reshape(xrec.shape)) ** 2)
    with np.errstate(divide='ignore'):
        return 10 * np.log10(1.0 / err)


def psnr(xrec, w):
    """reference problems/problem.py:33-35 (data_range = 1 for float images >= 0)."""
    return np.around(psnr_raw(xrec, w), decimals=2)


# --------------------------------------------------------------------------
# DnCNN-17 (torch CPU fp32: the floating-point reference of the MFMA kernel)
# --------------------------------------------------------------------------
def dncnn_forward(weights, x):
    """Plain conv/BN(eval)/ReLU stack of reference DeepDenoisers/model/models.py:5-22.
    `weights` = dict from tests/golden (keys conv{i}.weight, bn{i}.{weight,bias,mean,var});
    x: (H, W) float32 array.  Returns the predicted residual (H, W) float32."""
    import torch
    import torch.nn.functional as F
    n_layers = int(weights['n_layers'])
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))[None, None]
    with torch.no_grad():
        for i in range(n_layers):
            t = F.conv2d(t, torch.from_numpy(weights[f'conv{i}.weight']), padding=1)
            if f'bn{i}.weight' in weights:
                t = F.batch_norm(t, torch.from_numpy(weights[f'bn{i}.mean']),
                                 torch.from_numpy(weights[f'bn{i}.var']),
                                 torch.from_numpy(weights[f'bn{i}.weight']),
                                 torch.from_numpy(weights[f'bn{i}.bias']),
                                 training=False, eps=1e-5)
            if i < n_layers - 1:
                t = F.relu(t)
    return t[0, 0].numpy()


class DnCNNDenoiser:
    """reference denoisers/RealSN_DnCNN.py:16-42 around `dncnn_forward`."""

    def __init__(self, weights, sigma):
        self.t = 0
        self.sigma = sigma
        self.weights = weights

    def denoise(self, noisy, sigma_est=0):
        x = np.copy(noisy)
        lo, hi = np.min(x), np.max(x)
        x = (x - lo) / (hi - lo)
        scale_range = 1.0 + self.sigma / 255.0 / 2.0
        scale_shift = (1 - scale_range) / 2.0
        x = x * scale_range + scale_shift
        r = dncnn_forward(self.weights, x.astype(np.float32))
        x = x - r
        x = (x - scale_shift) / scale_range
        return x * (hi - lo) + lo


def mmo_forward(weights, x):
    """`simple_CNN.forward` of reference denoisers/MMODenoise.py:88-101 (depth-n conv stack with biases, LeakyReLU
    after all but the last conv, input added to the output).  x: 2-D float32 array."""
    import torch
    import torch.nn.functional as F
    n_layers = int(weights['n_layers'])
    slope = float(weights['negative_slope'])
    x_in = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))[None, None]
    t = x_in
    with torch.no_grad():
        for i in range(n_layers):
            t = F.conv2d(t, torch.from_numpy(weights[f'conv{i}.weight']), torch.from_numpy(weights[f'conv{i}.bias']), padding=1)
            if i < n_layers - 1:
                t = F.leaky_relu(t, slope)
        t = t + x_in
    return t[0, 0].numpy()


class MMODenoiser:
    """reference denoisers/MMODenoise.py:122-128 around apply_model (:18-40): the network sees the TRANSPOSED image
    (np.moveaxis(noisy, -1, 0) on a 2-D array), clamped to [0, 1] in fp32; the output is clamped, transposed back
    and clipped.  Returns float32 like the reference."""

    def __init__(self, weights):
        self.t = 0
        self.weights = weights

    def denoise(self, noisy, sigma_est=0):
        self.t += 1
        xt = np.clip(np.moveaxis(np.asarray(noisy), -1, 0).astype(np.float32), 0.0, 1.0)
        y = np.clip(mmo_forward(self.weights, xt), 0.0, 1.0)
        return np.clip(np.moveaxis(y, 0, -1), 0.0, 1.0)
