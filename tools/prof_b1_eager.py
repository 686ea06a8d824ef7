"""The drop-in loops as a user of the reference runs them (B = 1, wall clock, eager): inner iterations per second and a cProfile."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import problems as P, denoisers as D, algorithms as A
IMG = os.path.join(ROOT, 'tests', 'golden', 'synth256.png')
which = sys.argv[1] if len(sys.argv) > 1 else 'tv'
def mk_den():
    if which == 'tv':
        return D.TVDenoiser()
    if which == 'saga':
        d = D.NLMDenoiser(patch_size=5, patch_distance=5, dtype=torch.float32)
        d.sigma = 0.05
        return d
    W = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'dncnn_noise15.npz')))
    return D.RealSN_DnCNNDenoiser('RealSN_DnCNN', sigma=15, weights=W)
den = mk_den()
for rep, prof in ((0, False), (1, False), (2, True)):
    np.random.seed(0)
    if which == 'saga':
        p = P.Deblur(IMG, H=256, W=256, kernel='Minimal', scale_percent=100, snr=20., dtype=torch.float32)
    else:
        p = P.CSMRI(IMG, H=256, W=256, sample_prob=0.2, snr=20., dtype=torch.float32)
    np.random.seed(1)
    torch.cuda.synchronize()
    pr = cProfile.Profile() if prof else None
    if pr: pr.enable()
    t0 = time.perf_counter()
    if which == 'saga':
        r = A.pnp_saga(p, den, 0.5, 2.0, 3000, hist_size=50, verbose=False, converge_check=False)
    else:
        r = A.pnp_svrg(p, den, 2e3, 2.0, 10, 1000, verbose=False, converge_check=False, variant='svrg')
    dt = time.perf_counter() - t0
    if pr: pr.disable()
    n = len(r['time_per_iter'])
    print(f'{which}: {n} log entries in {dt:.2f} s = {dt / n * 1e6:.0f} us per entry', flush=True)
    if pr: pstats.Stats(pr).sort_stats('tottime').print_stats(18)
