"""The graph branch of the drop-in pnp_svrg (B = 1, TV prox) step by step, host / device time of each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import problems as P, denoisers as D, algorithms as A
from pnp_svrg_amd import algorithms as AA
IMG = os.path.join(ROOT, 'tests', 'golden', 'synth256.png')
n = 200; tt = 2 + 3 * (n // 10) + 5 * n - 1
def tm(tag, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{tag:46s} host {1e3 * (t1 - t0):8.3f} ms, + device {1e3 * (t2 - t1):8.3f} ms', flush=True)
    return r
for rep in range(3):
    print('--- rep', rep)
    np.random.seed(0)
    p = tm('P.CSMRI(...)', lambda: P.CSMRI(IMG, H=256, W=256, sample_prob=0.2, snr=20., dtype=torch.float32))
    np.random.seed(1)
    d = D.TVDenoiser(); clock = A.CountingClock()
    c = tm('_Ctx', lambda: AA._Ctx(p, d, clock))
    tm('eligible', lambda: AA._svrg_graph_eligible(c, p, d, clock, 1, False, False, False))
    outers = tm('schedule', lambda: AA._svrg_graph_schedule(c, p, tt, 10, 1000))
    n_log = 1 + sum(1 + len(l) for l in outers)
    run = tm('_SvrgGraph(...)', lambda: AA._SvrgGraph(p, d, 2e3, 10, 1000, 'svrg', n_log))
    tm('upload', lambda: run.upload(outers))
    tm('log_psnr', lambda: run.log_psnr())
    tm('run_outer(0) (capture)', lambda: run.run_outer(0, len(outers[0])))
    tm('run_outer(1..)', lambda: [run.run_outer(o, len(outers[o])) for o in range(1, len(outers))])
    sse = tm('log readback', lambda: run.log[:, 0].cpu().numpy())
    tm('psnr list', lambda: [p.psnr_from_sse(v, p.N) for v in sse])
    tm('result', lambda: c.result(run.z.reshape(-1), 'PnP SVRG'))
    t0 = time.perf_counter(); del run; torch.cuda.synchronize(); print(f'del run {1e3 * (time.perf_counter() - t0):.3f} ms')
