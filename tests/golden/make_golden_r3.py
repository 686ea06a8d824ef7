#!/opt/conda/bin/python3.9
"""Round-3 golden vectors: the measured configuration at its FULL length, produced by RUNNING THE REAL REFERENCE in the
build container (harness conventions of make_golden.py: modules imported by path, counting clock bound to the algorithm
module's `time` name, arrays only).

    /opt/conda/bin/python3.9 tests/golden/make_golden_r3.py

Writes tests/golden/traces256_full.npz -- SURVEY 8(d): np.random.seed(0); CSMRI(synth256.png, 256, 256, sample_prob=0.2,
snr=20.); np.random.seed(1); eta = 2e3, T2 = 10, mini_batch_size = 1000, lr_decay = 1, converge_check = False, 20 outer x 10
inner = 200 inner iterations, TV prox:
  * svrg_*      : algorithms/pnp_svrg.py as v1 executes it (v = mu, :54)
  * truesvrg_*  : the true SVRG direction (:53) composed by the harness from the reference's own grad_full / grad_stoch /
                  select_mb / estimate_sigma / TVDenoiser.denoise / PSNR calls, same loop nest and RNG draws
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
_pl.Identity = object
_pl.signalprocessing = types.SimpleNamespace()
sys.modules['pylops'] = _pl

import algorithms                                            # noqa: E402,F401
from CSMRI import CSMRI                                      # noqa: E402
from TV import TVDenoiser                                    # noqa: E402
from skimage.restoration import estimate_sigma               # noqa: E402

N_OUTER, T2, MB, ETA = 20, 10, 1000, 2e3


class FakeClock:
    def __init__(self):
        self.n = -1.0

    def time(self):
        self.n += 1.0
        return self.n


def true_svrg(p, d, eta, n_outer, T2, mb_size):
    z = np.copy(p.Xinit)
    ps = [p.PSNR(z)]
    for i in range(n_outer):
        mu = p.grad_full(z)
        w = np.copy(z)
        ps.append(p.PSNR(z))
        for _ in range(T2):
            mb = p.select_mb(mb_size)
            v = (p.grad_stoch(z, mb) - p.grad_stoch(w, mb)) / mb_size + mu
            z -= eta * v
            z0 = np.copy(z).reshape(p.H, p.W)
            z0 = d.denoise(noisy=z0, sigma_est=estimate_sigma(z0, multichannel=True, average_sigmas=True))
            ps.append(p.PSNR(z0))
            z = np.copy(z0).ravel()
    return z, np.array(ps)


def main():
    img = os.path.join(HERE, 'synth256.png')
    out = {}
    np.random.seed(0)
    p = CSMRI(img, H=256, W=256, sample_prob=0.2, snr=20.)
    np.random.seed(1)
    mod = sys.modules['algorithms.pnp_svrg']
    mod.time = FakeClock()
    # clock calls: 2 in the prologue, 3 per outer + 5 per inner iteration
    r = mod.pnp_svrg(p, TVDenoiser(), ETA, 2 + N_OUTER * (3 + 5 * T2), T2, MB, verbose=False, converge_check=False)
    out['svrg_z'], out['svrg_psnr'] = r['z'], np.array(r['psnr_per_iter'])
    assert len(out['svrg_psnr']) == 1 + N_OUTER * (T2 + 1), len(out['svrg_psnr'])
    np.random.seed(0)
    p = CSMRI(img, H=256, W=256, sample_prob=0.2, snr=20.)
    np.random.seed(1)
    out['truesvrg_z'], out['truesvrg_psnr'] = true_svrg(p, TVDenoiser(), ETA, N_OUTER, T2, MB)
    print('svrg', out['svrg_psnr'][[0, 1, -1]], len(out['svrg_psnr']), 'true', out['truesvrg_psnr'][[0, 1, -1]], len(out['truesvrg_psnr']))
    np.savez_compressed(os.path.join(HERE, 'traces256_full.npz'), **out)


if __name__ == '__main__':
    main()
