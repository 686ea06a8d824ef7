"""Size-independent properties at BASELINE's full size (256 x 256, batched), where the oracle is
only spot-checked: linearity / adjoint identities of the gradients, the identities of SURVEY
section 4, batch invariance of the network, and checksums of the error sums."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def batch():
    from pnp_svrg_amd.engine import CsmriBatch
    return CsmriBatch.synthetic(8, 256, 256, 0.2, 20.0, seed=3, dtype=torch.float64)


def test_grad_stoch_all_ones_equals_grad_full(batch):
    """grad_stoch(x, all-ones)/M0 == grad_full(x) for CSMRI."""
    p = batch.plan
    x = torch.rand(8, 256, 256, dtype=torch.float64, device='cuda')
    gf = p.grad(x, batch.maskT, yh=batch.yh_full, alpha_vec=batch.inv_m0)
    # the selector mask o ones is the mask itself, rebuilt from index lists
    # (Bernoulli masks: a different number of sampled points per problem -> padded index lists, -1 = no entry)
    lists = [np.flatnonzero(m) for m in batch.mask_np]
    idx = np.full((len(lists), max(len(l) for l in lists)), -1, np.int32)
    for i, l in enumerate(lists):
        idx[i, :len(l)] = l
    sel = p.sel_from_indices(torch.from_numpy(idx).cuda())
    assert torch.equal(sel, batch.maskT)
    gs = p.grad(x, sel, yh=batch.yh_full)
    assert (gs * batch.inv_m0[:, None, None] - gf).abs().max().item() < 1e-15
    assert len(set(batch.M0.tolist())) > 1


def test_f13_difference_identity_and_linearity(batch):
    """gs(a) - gs(b) == the fused difference call (Y terms cancel); linear in its input."""
    p = batch.plan
    a = torch.rand(8, 256, 256, dtype=torch.float64, device='cuda')
    b = torch.rand_like(a)
    idx = batch.draw_minibatches(1, 1000, seed=5)[0]
    sel = p.sel_from_indices(idx)
    zero_yh = torch.zeros_like(batch.yh_full)
    ga = p.grad(a, sel, yh=zero_yh)
    gb = p.grad(b, sel, yh=zero_yh)
    gd = p.grad(a, sel, b=b)
    assert (ga - gb - gd).abs().max().item() < 1e-12
    g2 = p.grad(2.5 * a - 0.5 * b, sel)
    assert (g2 - (2.5 * ga - 0.5 * gb)).abs().max().item() < 1e-11
    # the operator z -> Re ifft2(sel o fft2 z) is symmetric: <g(a), b> == <a, g(b)>
    lhs = (ga * b).sum(dim=(1, 2))
    rhs = (a * gb).sum(dim=(1, 2))
    assert ((lhs - rhs).abs() / lhs.abs()).max().item() < 1e-11


def test_masked_fft_is_projection_scaled(batch):
    """With the full k-space selected, Re ifft2(fft2 z) == z (round trip through all three kernels)."""
    p = batch.plan
    z = torch.rand(8, 256, 256, dtype=torch.float64, device='cuda')
    ones = torch.ones(8, 256, 256, dtype=torch.uint8, device='cuda')
    out = p.grad(z, ones)
    assert (out - z).abs().max().item() < 1e-13
    zf = z.float()
    from pnp_svrg_amd import ops
    p32 = ops.CsmriPlan(256, 256, 8, torch.float32)
    assert (p32.grad(zf, ones) - zf).abs().max().item() < 2e-6


def test_dncnn_batch_invariance_and_shift():
    """Each image of a batch gets the same result as alone; the net commutes with a circular shift in
    the interior (translation equivariance of a conv stack away from the zero-padded border)."""
    from pnp_svrg_amd import ops
    from pnp_svrg_amd.denoisers import random_dncnn_weights
    w = random_dncnn_weights(17, seed=4)
    x = torch.rand(5, 256, 256, device='cuda')
    r5 = ops.DncnnPlan(w, 256, 256, 5).forward(x)
    p1 = ops.DncnnPlan(w, 256, 256, 1)
    for i in (0, 4):
        assert torch.equal(p1.forward(x[i:i + 1])[0], r5[i])          # bit-identical: same arithmetic per tile
    xs = torch.roll(x[:1], shifts=(8, 32), dims=(1, 2))
    rs = p1.forward(xs)[0]
    ref = torch.roll(r5[0], shifts=(8, 32), dims=(0, 1))
    core = (slice(8 + 20, 256 - 20), slice(32 + 20, 256 - 20))        # 17 layers -> 17-pixel border influence
    assert (rs[core] - ref[core]).abs().max().item() < 2e-4


def test_prox_tv_batch_and_error_sum_checksum(batch):
    from pnp_svrg_amd import ops
    z = batch.xinit.clone()
    out, sse, sig = ops.prox_tv(z, xrec=batch.xrec)
    o1, s1, g1 = ops.prox_tv(z[3:4].contiguous(), xrec=batch.xrec[3:4].contiguous())
    assert torch.equal(o1[0], out[3]) and s1[0] == sse[3] and g1[0] == sig[3]
    total = ((batch.xrec - out) ** 2).sum().item()
    assert abs(sse.sum().item() - total) <= 1e-9 * total
    assert abs(ops.sse(out, batch.xrec).sum().item() - total) <= 1e-9 * total
