"""Where the drop-in pnp_svrg + TV loop (B = 1, deterministic clock, hipGraph replay) spends its wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import problems as P, denoisers as D, algorithms as A
from pnp_svrg_amd import algorithms as AA
IMG = os.path.join(ROOT, 'tests', 'golden', 'synth256.png')
def mk():
    np.random.seed(0)
    return P.CSMRI(IMG, H=256, W=256, sample_prob=0.2, snr=20., dtype=torch.float32)
n = 200
tt = 2 + 3 * (n // 10) + 5 * n - 1
for rep in range(2):
    p = mk(); np.random.seed(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    A.pnp_svrg(p, D.TVDenoiser(), 2e3, tt, 10, 1000, verbose=False, converge_check=False, clock=A.CountingClock(), variant='svrg')
    torch.cuda.synchronize(); print(f'rep {rep}: whole call {(time.perf_counter() - t0) / n * 1e6:.1f} us/inner', flush=True)
# host side alone: the schedule generator
p = mk(); np.random.seed(1)
c = AA._Ctx(p, D.TVDenoiser(), A.CountingClock())
t0 = time.perf_counter()
outers = list(AA._svrg_graph_schedule(c, p, tt, 10, 1000))
print(f'host schedule alone: {(time.perf_counter() - t0) / n * 1e6:.1f} us/inner ({len(outers)} outer iterations)', flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
run = AA._SvrgGraph(p, D.TVDenoiser(), 2e3, 10, 1000, 'svrg', 8192)
run.upload(outers)
run.log_psnr()
torch.cuda.synchronize(); t1 = time.perf_counter()
run.run_outer(0, len(outers[0])); torch.cuda.synchronize(); t2 = time.perf_counter()
for o in range(1, len(outers)):
    run.run_outer(o, len(outers[o]))
t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
print(f'setup + upload {1e3 * (t1 - t0):.2f} ms; first outer iteration (capture + instantiate + replay) {1e3 * (t2 - t1):.2f} ms; '
      f'{len(outers) - 1} replays: host {(t3 - t2) / (len(outers) - 1) * 1e3:.3f} ms per outer, device done after {1e3 * (t4 - t2):.2f} ms '
      f'= {1e6 * (t4 - t2) / (n - 10):.1f} us per inner iteration', flush=True)
