"""cProfile of one drop-in pnp_svrg + TV call at B = 1 (deterministic clock, hipGraph replay): where the host time goes."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import problems as P, denoisers as D, algorithms as A
IMG = os.path.join(ROOT, 'tests', 'golden', 'synth256.png')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tt = 2 + 3 * (n // 10) + 5 * n - 1
def run(profile):
    np.random.seed(0)
    p = P.CSMRI(IMG, H=256, W=256, sample_prob=0.2, snr=20., dtype=torch.float32)
    np.random.seed(1)
    torch.cuda.synchronize()
    pr = cProfile.Profile() if profile else None
    t0 = time.perf_counter()
    if pr: pr.enable()
    A.pnp_svrg(p, D.TVDenoiser(), 2e3, tt, 10, 1000, verbose=False, converge_check=False, clock=A.CountingClock(), variant='svrg')
    torch.cuda.synchronize()
    if pr: pr.disable()
    print(f'whole call {(time.perf_counter() - t0) * 1e3:.1f} ms = {(time.perf_counter() - t0) / n * 1e6:.1f} us/inner', flush=True)
    if pr: pstats.Stats(pr).sort_stats('cumtime').print_stats(22)
run(False); run(False); run(True)
