"""Same-box A/B timing of library builds (tools/ab_build.sh): forward time of the DnCNN-17 plan at B = 120, interleaved rounds."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from pnp_svrg_amd import ops
W = dict(np.load(os.path.join(%r, 'tests/golden/dncnn_noise15.npz')))
B = 120
plan = ops.DncnnPlan(W, 256, 256, B, winograd=5)
x = torch.rand(B, 256, 256, device='cuda'); out = torch.empty_like(x)
for _ in range(5): plan.forward(x, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): plan.forward(x, out)
e1.record(); torch.cuda.synchronize()
print('%%.3f' %% (e0.elapsed_time(e1) / 20))
''' % (ROOT, ROOT)
names = sys.argv[1:]
res = {n: [] for n in names}
for rnd in range(3):
    for n in names:
        env = dict(os.environ, PNP_HIP_LIB=os.path.join(ROOT, 'pnp_svrg_amd', 'lib', 'ab', n + '.so'))
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=300)
        res[n].append(float(out.stdout.strip().splitlines()[-1]))
for n in names:
    print(n, ' '.join('%.3f' % v for v in res[n]), 'ms/forward  min %.3f' % min(res[n]), flush=True)
