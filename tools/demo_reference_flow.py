"""The flow of reference pnp_csmri.py (lines 11-28) on the drop-in packages with a REAL clock: build a CSMRI
problem, run pnp_gd / pnp_sgd / pnp_svrg for a wall-clock budget with the DnCNN prox, print iterations and PSNR.
Uses the committed fixtures (synthetic image, DnCNN sigma=15 weights) because the reference's data/ and the
RealSN_DnCNN checkpoints are not shipped here."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from denoisers import *      # noqa: E402,F401,F403
from problems import *       # noqa: E402,F401,F403
from algorithms import *     # noqa: E402,F401,F403

np.random.seed(0)
main_problem = CSMRI(os.path.join(ROOT, 'tests/golden/synth256.png'), H=256, W=256, sample_prob=0.5, snr=30.)
wts = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
CNNDenoiser = RealSN_DnCNNDenoiser(model_type="DnCNN", sigma=15, weights=wts)
tmp = CNNDenoiser.denoise(main_problem.Xrec)
tt = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
for name, run in (('pnp_gd', lambda: pnp_gd(main_problem, denoiser=CNNDenoiser, eta=1e3, tt=tt, verbose=False, converge_check=False)),
                  ('pnp_sgd', lambda: pnp_sgd(main_problem, denoiser=CNNDenoiser, eta=1e3, tt=tt, mini_batch_size=main_problem.M0, verbose=False, converge_check=False)),
                  ('pnp_svrg', lambda: pnp_svrg(main_problem, denoiser=CNNDenoiser, eta=1e3, tt=tt, T2=1, mini_batch_size=main_problem.M0, verbose=False, converge_check=False))):
    t0 = time.time()
    r = run()
    n = len(r['psnr_per_iter'])
    print(f"{name:9s} {r['algo_name']:9s}: {n:5d} log entries in {time.time() - t0:.2f} s  PSNR {r['psnr_per_iter'][0]} -> {r['psnr_per_iter'][-1]}"
          f"  (gradient {r['gradient_time']:.2f} s, denoise {r['denoise_time']:.2f} s)")
