"""One-block-row form of the F(4x4,3x3) kernel (PNP_W44_ROWS=1) against the two-row form and the direct kernel; B = 1 timing."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from pnp_svrg_amd import ops
W = dict(np.load(os.path.join(%r, 'tests/golden/dncnn_noise15.npz')))
rng = np.random.default_rng(0)
res = {}
for (H, Wd, B) in ((64, 64, 1), (256, 256, 1), (72, 128, 3), (256, 256, 3)):
    x = torch.from_numpy(rng.random((B, H, Wd)).astype(np.float32)).cuda()
    r5 = ops.DncnnPlan(W, H, Wd, B, winograd=5).forward(x).cpu().numpy()
    np.save('/tmp/w44rows_%%s_%%d_%%d_%%d.npy' %% (os.environ.get('PNP_W44_ROWS', 'auto'), H, Wd, B), r5)
    r0 = ops.DncnnPlan(W, H, Wd, B, winograd=0).forward(x).cpu().numpy()
    print(os.environ.get('PNP_W44_ROWS', 'auto'), H, Wd, B, 'max |w44 - direct| = %%.3e' %% np.abs(r5 - r0).max(), flush=True)
plan = ops.DncnnPlan(W, 256, 256, 1, winograd=5)
x = torch.rand(1, 256, 256, device='cuda'); out = torch.empty_like(x)
for _ in range(5): plan.forward(x, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): plan.forward(x, out)
e1.record(); torch.cuda.synchronize()
print(os.environ.get('PNP_W44_ROWS', 'auto'), 'B=1 forward %%.1f us' %% (e0.elapsed_time(e1) / 50 * 1e3), flush=True)
''' % (ROOT, ROOT)
import numpy as np
for rows in ('1', '2'):
    out = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, PNP_W44_ROWS=rows), capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr[-500:] if out.returncode else '')
for shp in ((64, 64, 1), (256, 256, 1), (72, 128, 3), (256, 256, 3)):
    a = np.load('/tmp/w44rows_1_%d_%d_%d.npy' % shp); b = np.load('/tmp/w44rows_2_%d_%d_%d.npy' % shp)
    print(shp, 'one-row == two-row bit for bit:', np.array_equal(a, b), ' max diff %.2e' % np.abs(a - b).max())
