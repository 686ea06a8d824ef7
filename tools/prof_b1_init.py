"""Where the 80 ms of _SvrgGraph.__init__ go (B = 1 drop-in call): timed pieces, GPU idle before each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import problems as P, denoisers as D, algorithms as A
IMG = os.path.join(ROOT, 'tests', 'golden', 'synth256.png')
def tm(tag, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{tag:46s} host {1e3 * (t1 - t0):8.3f} ms, + device {1e3 * (t2 - t1):8.3f} ms', flush=True)
    return r
for rep in range(3):
    print('--- rep', rep)
    np.random.seed(0)
    p = tm('P.CSMRI(...)', lambda: P.CSMRI(IMG, H=256, W=256, sample_prob=0.2, snr=20., dtype=torch.float32))
    tm('type(p.Xinit), first access', lambda: p.Xinit)
    x64 = tm('np.ascontiguousarray(p.Xinit, float64)', lambda: np.ascontiguousarray(p.Xinit, dtype=np.float64))
    print('   ', type(p.Xinit), getattr(p.Xinit, 'dtype', None), getattr(p.Xinit, 'shape', None), x64.flags['C_CONTIGUOUS'])
    t = tm('torch.from_numpy(x64)', lambda: torch.from_numpy(x64))
    tm('.to(cuda, float32)', lambda: t.to(device=p.device, dtype=p.dtype))
    tm('.to(cuda, float32) again', lambda: t.to(device=p.device, dtype=p.dtype))
    tm('p.to_device(p.Xinit)', lambda: p.to_device(p.Xinit))
    tm('torch.zeros((2051, 1), f64, cuda)', lambda: torch.zeros((2051, 1), dtype=torch.float64, device=p.device))
print('--- after 0.3 s of sleep each')
x64 = np.ascontiguousarray(p.Xinit, dtype=np.float64); t = torch.from_numpy(x64)
time.sleep(0.3); tm('.to(cuda, float32)', lambda: t.to(device=p.device, dtype=p.dtype))
time.sleep(0.3); tm('numpy cast + .to(cuda)', lambda: torch.from_numpy(x64.astype(np.float32)).to(device=p.device))
time.sleep(0.3); tm('.to(cuda, float32)', lambda: t.to(device=p.device, dtype=p.dtype))
print('--- after 200 host-only draws')
from pnp_svrg_amd import legacy_rng
for _ in range(200): legacy_rng.choice(p._mask_locs if hasattr(p, '_mask_locs') else 13107, 1000)
tm('.to(cuda, float32)', lambda: t.to(device=p.device, dtype=p.dtype))
for _ in range(200): np.random.choice(13107, 1000, replace=False)
tm('.to(cuda, float32) after numpy draws', lambda: t.to(device=p.device, dtype=p.dtype))
