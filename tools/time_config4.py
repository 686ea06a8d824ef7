"""Timing of the config-4 kernels (Deblur gradient, NLM prox) and the PR gradient at BASELINE sizes."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnp_svrg_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

rng = np.random.default_rng(0)
for dtype in (torch.float32, torch.float64):
    for B in (1, 16, 64):
        z = torch.rand(B, 256, 256, dtype=dtype, device='cuda')
        xrec = torch.rand_like(z)
        sig = torch.full((B,), 0.05, dtype=dtype, device='cuda')
        ms = timeit(lambda: ops.nlm2d(z, sigma_in=sig, xrec=xrec))
        print(f'NLM 256^2 {str(dtype)[6:]} B={B}: {ms:.3f} ms  ({ms/B*1e3:.1f} us/image; ~{65536*121*25*5*B/ms/1e9:.1f} GFLOP/s nominal)')
    Bk = np.zeros(65536); Bk[[0, 32832, 32853, 32896]] = 0.25 / 65536
    for B in (1, 64):
        plan = ops.DeblurPlan(256, 256, B, dtype, Bk)
        z = torch.rand(B, 65536, dtype=dtype, device='cuda'); Y = torch.rand_like(z); out = torch.empty_like(z)
        ms = timeit(lambda: plan.grad(z, Y, scale=1.0, out=out))
        print(f'Deblur grad 256^2 {str(dtype)[6:]} B={B}: {ms:.3f} ms ({ms/B*1e3:.1f} us/problem)')
M, N = 8192, 16384
A = torch.randn(M, N, device='cuda'); w = torch.rand(N, device='cuda'); y = torch.rand(M, device='cuda')
ws = None
ms = timeit(lambda: ops.pr_grad(A, w, y, scale=1.0 / M), n=5)
print(f'PR grad_full 8192x16384 f32: {ms:.3f} ms -> {2*M*N*4/ms/1e6:.0f} GB/s over A')
rows = torch.from_numpy(rng.choice(M, 800, replace=False).astype(np.int32)).cuda()
ms = timeit(lambda: ops.pr_grad(A, w, y, rows=rows), n=20)
print(f'PR grad_stoch (800 rows): {ms:.3f} ms -> {2*800*N*4/ms/1e6:.0f} GB/s over A rows')
# batched form (PrBatch): B problems, each with its own A (128 x 128 images, alpha = 0.5 as in the reference's PR notebook)
B = 4
Ab = torch.randn(B, M, N, device='cuda'); wb = torch.rand(B, N, device='cuda'); yb = torch.rand(B, M, device='cuda')
ms = timeit(lambda: ops.pr_grad_batch(Ab, wb, yb, scale=1.0 / M), n=5)
print(f'PR grad_full batched B={B} 8192x16384 f32: {ms:.3f} ms -> {2*B*M*N*4/ms/1e6:.0f} GB/s over A')
# SAGA table update (pnp_saga_table_update): 8 vectors of B x 65536 floats
for B in (64, 256):
    z, g, slot, prev, ts = (torch.rand(B, 65536, device='cuda') for _ in range(5))
    ms = timeit(lambda: ops.saga_table_update(z, g, slot, prev, ts, 0.1, 0.02), n=20)
    print(f'SAGA table update B={B}: {ms*1e3:.1f} us -> {8*B*65536*4/ms/1e6:.0f} GB/s (8 vector passes)')
