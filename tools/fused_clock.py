"""Steady-state phase timing of the one-kernel iteration from in-kernel clock stamps (diagnostic build):

    tools/ab_tree.sh clk "sed -i 's/^FLAGS := /FLAGS := -DPNP_FUSED_CLOCK /' Makefile"
    PNP_HIP_LIB=pnp_svrg_amd/lib/ab/clk.so python tools/fused_clock.py [B]

Thread 0 of every workgroup stamps the shader clock at the phase boundaries of k_svrg_iter (csrc/csmri_fused.hip, PNP_STAMP);
with B = 1024 the four workgroups a CU runs one after the other are out of step with the other CUs', which is the state the
bench measures (the PNP_FUSED_STOP builds time every CU in the same phase at once)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pnp_svrg_amd import _native as N
from pnp_svrg_amd.engine import CsmriBatch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = CsmriBatch.synthetic(B, 256, 256, 0.2, 20.0, seed=1)
p = b.plan
z, w, mu = b.xinit.clone(), b.xinit.clone() * 0.9, b.xinit.clone() * 1e-4
sel = torch.empty((1, B, 256, 8), dtype=torch.int32, device='cuda')
p.draw_thresholds(b.bits, 1000, 1, 0, 1, selbits=sel)
sse = torch.empty(B, dtype=torch.float64, device='cuda')
out = torch.empty_like(z)
for _ in range(5):
    p.svrg_step(z, w, sel[0], alpha=-2.0, beta=1.0, c1=z, gamma=-2e3, c2=mu, out=out, xrec=b.xrec, sse=sse)
torch.cuda.synchronize()
lib = N.lib()
nb = min(B, 4096)
buf = (ctypes.c_ulonglong * (nb * 16))()
lib.pnp_debug_fused_stamps.restype = ctypes.c_int
lib.pnp_debug_fused_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.pnp_debug_fused_stamps(buf, nb) == 0
st_all = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 16)
st = st_all[:, :10].astype(np.int64)
d = np.diff(st, axis=1).astype(np.float64)
names = ['1 operand loads a, b (+ twiddles, selector bits)', '1 rows forward FFT', '2 columns (two halves)', '3 rows inverse FFT',
         '3 epilogue operands c1, c2', '4 re-layout', '5 noise estimate (median)', '5 Haar BayesShrink', '5 error + store']
tot = (st[:, 9] - st[:, 0]).astype(np.float64)
order = np.argsort(st[:, 0])
late = order[len(order) // 2:]            # workgroups that started after the first wave of the launch
print(f'B = {B}: shader cycles per workgroup, median over all / over the later half of the launch (steady state)')
for i, n in enumerate(names):
    print(f'  {n:55s} {np.median(d[:, i]):9.0f} {np.median(d[late, i]):9.0f}   {100 * np.median(d[late, i]) / np.median(tot[late]):5.1f} %')
print(f'  {"whole workgroup":55s} {np.median(tot):9.0f} {np.median(tot[late]):9.0f}')
if st_all[:, 10].any():
    fine = st_all[:, [8, 10, 11, 12, 13, 9]].astype(np.int64)
    df = np.diff(fine, axis=1).astype(np.float64)
    for nm, i in (("5a x[0] -> buffer, ground truth of half 0 requested and waited for", 0), ("5b half 0: pieces read, error, stores issued", 1),
                  ("5c x[1] -> buffer, wait for the ground truth of half 1", 2), ("5d half 1: pieces read, error, stores issued", 3),
                  ("5e error reduction, stores drained", 4)):
        print(f'     {nm:72s} {np.median(df[late, i]):9.0f}')
# memory phases as effective bandwidth per CU (bytes / cycle): P1 reads a, b (512 KB); the epilogue c1, c2 (512 KB); the last
# phase reads the ground truth and writes the result (512 KB)
for nm, i in (('operand loads', 0), ('epilogue operands', 4), ('error + store', 8)):
    print(f'  {nm:20s}: {524288 / np.median(d[late, i]):6.1f} B/clk per CU')
# start times of the workgroups relative to the first (per round of 256): does a start-up stagger persist?
s0 = np.sort(st[:, 0] - st[:, 0].min())
for r in range(0, nb, 256):
    blk = s0[r:r + 256]
    print(f'  workgroups {r:4d}..{r + len(blk) - 1:4d} by start time: first {blk[0]:9d}  median {int(np.median(blk)):9d}  last {blk[-1]:9d}  (spread {blk[-1] - blk[0]})')
