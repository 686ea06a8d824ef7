"""NLM prox: register-strip kernel (default at search radius 5) against the LDS-streaming form (PNP_NLM_GENERIC=1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pnp_svrg_amd import ops
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for dtype in (torch.float32, torch.float64):
    for B in (1, 64):
        z = torch.rand(B, 256, 256, dtype=dtype, device='cuda') * 0.2 + 0.4
        xrec = torch.rand_like(z)
        sig = torch.full((B,), 0.05, dtype=dtype, device='cuda')
        ms = timeit(lambda: ops.nlm2d(z, sigma_in=sig, xrec=xrec))
        print(f'{"generic" if os.environ.get("PNP_NLM_GENERIC") else "strip"} NLM 256^2 {str(dtype)[6:]} B={B}: {ms:.3f} ms  ({ms/B*1e3:.1f} us/image; {65536*121*25*5*B/ms/1e9/1e3:.1f} nominal TFLOP/s)', flush=True)
