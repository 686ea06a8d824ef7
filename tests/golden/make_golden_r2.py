#!/opt/conda/bin/python3.9
"""Round-2 golden vectors, again produced by RUNNING THE REAL REFERENCE in the build container
(see make_golden.py for the harness conventions: modules imported by path, counting clock, pylops Identity shim,
arrays only).

    /opt/conda/bin/python3.9 tests/golden/make_golden_r2.py

Writes tests/golden/r2_fixtures.npz:
  * resize_*      : the pixels of the reference's data/Set12/08.png (512 x 512, a data file) and the Xrec the reference's
                    Problem.__init__ makes of it at 256 x 256 (PIL bicubic resize + min-max, problems/problem.py:16-24)
  * kernelpng_*   : the pixels of data/kernel.png and the blur vector B that Deblur(kernel_path=...) builds from it at
                    64 x 64 and 256 x 256 (problems/DeblurSR.py:72-78,93)
  * nlm256_*      : denoise_nl_means (slow mode) on a 256 x 256 iterate, through NLMDenoiser.denoise
  * c4_*          : BASELINE config 4 at full size: Deblur 256 x 256 ("Minimal" kernel, scale 100 %), NLM prox, pnp_saga
                    trace (counting clock) + the gradients at Xinit
"""
import os
import sys
import types
import warnings
import numpy as np

warnings.filterwarnings('ignore')
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [REF, REF + '/problems', REF + '/denoisers']
_pl = types.ModuleType('pylops')


class _Identity:
    def __init__(self, n):
        self.n = n
        self.H = self

    def __mul__(self, x):
        return x


_pl.Identity = _Identity
_pl.signalprocessing = types.SimpleNamespace()
sys.modules['pylops'] = _pl

import algorithms                                            # noqa: E402
from problem import Problem                                  # noqa: E402
from DeblurSR import Deblur                                  # noqa: E402
from NLM import NLMDenoiser                                  # noqa: E402
from skimage.restoration import estimate_sigma               # noqa: E402
from PIL import Image                                        # noqa: E402


class FakeClock:
    def __init__(self):
        self.n = -1.0

    def time(self):
        self.n += 1.0
        return self.n


def run_algo(name, *args, **kw):
    mod = sys.modules['algorithms.' + name]
    mod.time = FakeClock()
    return getattr(mod, name)(*args, verbose=False, **kw)


def main():
    out = {}
    # ---- a6: PIL bicubic resize of a 512^2 image
    src = REF + '/data/Set12/08.png'
    out['resize_pixels'] = np.array(Image.open(src))
    p = Problem(src, 256, 256)
    out['resize_Xrec'] = p.Xrec
    # ---- a13: kernel_path branch
    kp = REF + '/data/kernel.png'
    out['kernelpng_pixels'] = np.array(Image.open(kp))
    img64, img256 = os.path.join(HERE, 'synth64.png'), os.path.join(HERE, 'synth256.png')
    np.random.seed(0)
    p = Deblur(img64, H=64, W=64, kernel_path=kp, scale_percent=100, snr=20.)
    out['kernelpng_B64'] = p.B
    np.random.seed(0)
    p = Deblur(img256, H=256, W=256, kernel_path=kp, scale_percent=100, snr=20.)
    out['kernelpng_B256'] = p.B
    out['kernelpng_sigma256'] = p.sigma
    out['kernelpng_grad_full256'] = p.grad_full(p.Xinit)
    out['kernelpng_Xinit256'] = p.Xinit
    out['kernelpng_Y256'] = p.Y
    # ---- a19 at 256^2
    den = np.load(os.path.join(HERE, 'denoise.npz'))
    z0 = den['r256_z0']
    s = estimate_sigma(z0, multichannel=True, average_sigmas=True)
    d = NLMDenoiser()
    d.sigma = 1.0
    out['nlm256_sigma_est'] = s
    out['nlm256_out'] = d.denoise(noisy=z0, sigma_est=s)
    # ---- config 4, full size
    np.random.seed(0)
    p = Deblur(img256, H=256, W=256, kernel='Minimal', scale_percent=100, snr=20.)
    np.random.seed(3)
    mb = p.select_mb(3000)
    out.update({'c4_Y': p.Y, 'c4_Xinit': p.Xinit, 'c4_sigma': p.sigma, 'c4_mb': mb.astype(np.uint8),
                'c4_grad_full': p.grad_full(p.Xinit), 'c4_grad_stoch': p.grad_stoch(p.Xinit, mb)})
    np.random.seed(1)
    d = NLMDenoiser()
    d.sigma = 1.0
    # clock: 3 in the prologue, 5 per iteration -> tt = 5 * steps - 1 gives `steps` iterations
    # the reference's blur has gain 1/sqrt(N) (B = kernel / N, fft_blur x sqrt(N)), so gradients are O(1e-9): eta = 1e9 makes
    # the data term move the iterate (a trace that is sensitive to the gradient, which is what pins parity)
    r = run_algo('pnp_saga', p, d, 1e9, 5 * 6 - 1, 3000, hist_size=4, converge_check=False)
    out['c4_saga_nlm_z'], out['c4_saga_nlm_psnr'] = r['z'], np.array(r['psnr_per_iter'])
    print('config 4 trace', r['psnr_per_iter'])
    np.savez_compressed(os.path.join(HERE, 'r2_fixtures.npz'), **out)
    print({k: getattr(v, 'shape', None) for k, v in out.items()})


if __name__ == '__main__':
    main()
