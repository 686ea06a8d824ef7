"""Determinism soak of the F(4x4,3x3) conv kernels (argv[1]: conv mode 5 or 6): repeated forward passes on the same input must be bit-identical (a stale
accumulator copy, a missed DMA wait or an LDS race would show up as run-to-run differences), across batch sizes that use
both region forms, full and partial waves of regions per CU, and under concurrent memory pressure from a second stream."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pnp_svrg_amd import ops
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 5           # 5 = the fp32 kernel, 6 = the 3 x bf16 split form
W = dict(np.load(os.path.join(ROOT, 'tests/golden/dncnn_noise15.npz')))
bad = 0
side = torch.cuda.Stream()
junk = torch.empty(64 * 1024 * 1024, device='cuda')
for B in (1, 2, 3, 7, 16, 33, 120):
    plan = ops.DncnnPlan(W, 256, 256, B, winograd=MODE)
    ref_plan = ops.DncnnPlan(W, 256, 256, B, winograd=0)
    x = torch.rand(B, 256, 256, device='cuda')
    first = plan.forward(x).clone()
    direct = ref_plan.forward(x)
    err = (first - direct).abs().max().item()
    n_diff = 0
    reps = 40 if B <= 16 else 12
    for i in range(reps):
        if i % 2:
            with torch.cuda.stream(side):
                junk.mul_(1.0001)                                    # HBM traffic from another stream while the conv runs
        out = plan.forward(x)
        if not torch.equal(out, first):
            n_diff += 1
    torch.cuda.synchronize()
    bad += n_diff
    print(f'B={B:4d}: {reps} repeats, {n_diff} differing from the first; max |mode {MODE} - direct| = {err:.2e}', flush=True)
print('SOAK', 'FAILED' if bad else 'ok')
sys.exit(1 if bad else 0)
