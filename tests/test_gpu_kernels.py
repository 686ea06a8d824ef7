"""GPU parity of the HIP kernels (through the C ABI) against the oracle on the same inputs
and against the committed golden vectors.  Tolerances are stated per test:
f64 path: <= 1e-11 absolute (rounding-order differences only); f32 path: relative to the
magnitude of the quantity, ~1e-5."""
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN

from oracle import denoise as od, problems as op

pytestmark = pytest.mark.gpu
IMG256 = os.path.join(GOLDEN, 'synth256.png')
IMG64 = os.path.join(GOLDEN, 'synth64.png')


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def csmri(path, n, seed=0):
    np.random.seed(seed)
    return op.CSMRI(path, H=n, W=n, sample_prob=0.2, snr=20.)


@pytest.fixture(scope='module')
def ops():
    from pnp_svrg_amd import ops as o
    o.require_gpu()
    return o


TOL = {torch.float64: 1e-11, torch.float32: 2e-5}


@pytest.mark.parametrize('n,path', [(256, IMG256), (64, IMG64)])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_csmri_grad(ops, g_csmri, n, path, dtype):
    tag = f's{n}'
    p = csmri(path, n)
    cdt = torch.complex128 if dtype == torch.float64 else torch.complex64
    B = 3                                    # batch of 3: the problem, a flipped copy, zeros
    plan = ops.CsmriPlan(n, n, B, dtype)
    z = np.stack([p.Xinit, p.Xinit[::-1].copy(), np.zeros(p.N)]).reshape(B, n, n)
    maskT = dev(np.ascontiguousarray(np.broadcast_to(p.mask.T.astype(np.uint8), (B, n, n))))
    YT = dev(np.ascontiguousarray(np.broadcast_to(p.Y.T, (B, n, n))), cdt)
    # selector built on device from index lists == dense transpose
    idx = dev(np.broadcast_to(np.flatnonzero(p.mask).astype(np.int32), (B, p.M0)).copy())
    selT = plan.sel_from_indices(idx)
    assert torch.equal(selT, maskT)
    dense = dev(np.ascontiguousarray(np.broadcast_to(p.mask.astype(np.uint8), (B, n, n))))
    assert torch.equal(plan.sel_from_dense(dense), maskT)
    yh = plan.pack_y(YT, selT)
    g = plan.grad(dev(z, dtype), selT, yh=yh, alpha=1.0 / p.M0).cpu().numpy().astype(np.float64)
    ref = np.stack([p.grad_full(zz.ravel()) for zz in z]).reshape(B, n, n)
    scale = np.abs(ref).max()
    assert np.abs(g - ref).max() <= TOL[dtype] * scale * 10
    # golden vector from the reference itself
    assert np.abs(g[0].ravel() - g_csmri[f'{tag}_grad_full']).max() <= TOL[dtype] * scale * 10

    # stochastic gradient with a minibatch selector (un-normalised, CSMRI.py:83-89)
    mb = g_csmri[f'{tag}_mb']
    sel = (p.mask * mb).astype(np.uint8)
    selT2 = plan.sel_from_dense(dev(np.ascontiguousarray(np.broadcast_to(sel, (B, n, n)))))
    yh2 = plan.pack_y(YT, selT2)
    gs = plan.grad(dev(z, dtype), selT2, yh=yh2).cpu().numpy().astype(np.float64)
    ref_s = g_csmri[f'{tag}_grad_stoch'].reshape(n, n)
    assert np.abs(gs[0] - ref_s).max() <= TOL[dtype] * np.abs(ref_s).max() * 10

    # SVRG correction fused with the step: z - lr*((gs(z)-gs(w))/mb + mu)
    w = z + 0.01 * np.cos(np.arange(B * n * n)).reshape(B, n, n)
    mu = ref
    lr, mbs = 2e3, int(mb.sum())
    zz = dev(z, dtype)
    out = plan.grad(zz, selT2, b=dev(w, dtype), alpha=-lr / mbs, beta=1.0, c1=zz, gamma=-lr, c2=dev(mu, dtype))
    exp = np.stack([z[i].ravel() - lr * ((p.grad_stoch(z[i].ravel(), mb) - p.grad_stoch(w[i].ravel(), mb)) / mbs
                                         + mu[i].ravel()) for i in range(B)]).reshape(B, n, n)
    assert np.abs(out.cpu().numpy() - exp).max() <= TOL[dtype] * 50


@pytest.mark.parametrize('tag,n', [('s256', 256), ('r256', 256), ('s64', 64)])
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_sigma_est_and_tv(ops, g_denoise, tag, n, dtype):
    g = g_denoise
    z0 = g[f'{tag}_z0']
    rng = np.random.default_rng(1)
    zb = np.stack([z0, z0[::-1].copy(), z0 + 0.05 * rng.standard_normal(z0.shape)])
    zt = dev(zb, dtype)
    s = ops.sigma_est(zt).cpu().numpy().astype(np.float64)
    s_ref = np.array([od.estimate_sigma(x) for x in zb])
    rel = 1e-12 if dtype == torch.float64 else 3e-5
    np.testing.assert_allclose(s, s_ref, rtol=rel)
    assert abs(s[0] - float(g[f'{tag}_sigma_est'])) <= rel * s_ref[0] * 2

    xrec = dev(np.clip(zb, 0, 1), dtype)
    out, sse, sig = ops.prox_tv(zt, xrec=xrec)
    ref = np.stack([od.haar_bayes_cols(x, sg) for x, sg in zip(zb, s_ref)])
    atol = 1e-12 if dtype == torch.float64 else 2e-5
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=0, atol=atol)
    np.testing.assert_allclose(out[0].cpu().numpy(), g[f'{tag}_tv'], rtol=0, atol=atol)
    np.testing.assert_allclose(sig.cpu().numpy(), s_ref, rtol=rel)
    sse_ref = ((np.clip(zb, 0, 1) - ref) ** 2).reshape(3, -1).sum(1)
    np.testing.assert_allclose(sse.cpu().numpy(), sse_ref, rtol=1e-10 if dtype == torch.float64 else 1e-4)
    np.testing.assert_allclose(ops.sse(out, xrec).cpu().numpy(), sse_ref, rtol=1e-10 if dtype == torch.float64 else 1e-4)

    # sigma_modifier, and the sigma_est <= 0 fallback branch (TV.py:23-26)
    out2, _, _ = ops.prox_tv(zt, sigma_modifier=1.7)
    np.testing.assert_allclose(out2[0].cpu().numpy(), g[f'{tag}_tv_mod'], rtol=0, atol=atol)
    zero_sig = torch.zeros(3, dtype=dtype, device='cuda')
    out3, _, _ = ops.prox_tv(zt, sigma_in=zero_sig, fallback_sigma=0.07 * 0.9)
    np.testing.assert_allclose(out3[0].cpu().numpy(), g[f'{tag}_tv_strength'], rtol=0, atol=atol)


def test_tv_edges(ops, g_denoise):
    g = g_denoise
    # sigma = 0 with exactly-zero detail coefficients: 0/0 -> NaN in pywt's soft threshold; same here
    zin = dev(g['edge_tv_sigma0_in'][None], torch.float64)
    out, _, _ = ops.prox_tv(zin, sigma_in=torch.zeros(1, dtype=torch.float64, device='cuda'), fallback_sigma=0.0)
    ref = g['edge_tv_sigma0']
    o = out[0].cpu().numpy()
    assert np.array_equal(np.isnan(o), np.isnan(ref))
    np.testing.assert_allclose(o[~np.isnan(ref)], ref[~np.isnan(ref)], atol=1e-14)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_median_even_and_zero_counts(ops, dtype):
    """Columns whose detail coefficients contain exact zeros (masked out of the median) give
    even and odd counts; constant columns give an empty set (NaN, like np.median([]))."""
    rng = np.random.default_rng(3)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    z = rng.random((4, 64, 64)).astype(npdt)
    z[0, 10:30, :] = 0.5                       # flat band -> exact-zero coefficients
    z[1, :, ::2] = 0.25                        # constant columns -> no nonzero coefficient -> NaN
    z[2, 20:24, 5] = 0.0
    s = ops.sigma_est(dev(z, dtype)).cpu().numpy()
    ref = np.array([od.estimate_sigma(x) for x in z])
    if dtype == torch.float64:                 # zero / non-zero classification is bit-exact only in f64
        assert np.isnan(s[1]) == np.isnan(ref[1])
    ok = ~np.isnan(ref)
    ok[1] = False
    np.testing.assert_allclose(s[ok], ref[ok], rtol=1e-12 if dtype == torch.float64 else 1e-4)


def test_minmax_axpby(ops):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5, 4096))
    t = dev(x, torch.float32)
    mm = ops.minmax(t).cpu().numpy()
    np.testing.assert_array_equal(mm[:, 0], x.astype(np.float32).min(1))
    np.testing.assert_array_equal(mm[:, 1], x.astype(np.float32).max(1))
    y = dev(rng.standard_normal((5, 4096)), torch.float64)
    t64 = dev(x, torch.float64)
    r = ops.axpbypcz(2.0, t64, -0.5, y, 0.25, t64)
    np.testing.assert_allclose(r.cpu().numpy(), 2.25 * x - 0.5 * y.cpu().numpy(), atol=1e-14)


@pytest.mark.parametrize('n,dt', [(64, torch.float64), (128, torch.float64), (256, torch.float64), (256, torch.float32)])
def test_selector_forms_agree(ops, n, dt):
    """The selector forms of the column pass give the same gradient: explicit uint8 selector, bit-packed mask, and
    mask o device-drawn minibatch as the draw kernel emits it (== the selector materialised from the threshold).  The
    data term formed inside the column pass from YT == the pre-packed one (pnp_csmri_pack_y).  Per-problem alpha_vec
    == per-problem scaling afterwards."""
    rng = np.random.default_rng(n)
    B = 3
    cdt = torch.complex128 if dt == torch.float64 else torch.complex64
    tol = 1e-12 if dt == torch.float64 else 2e-5
    plan = ops.CsmriPlan(n, n, B, dt)
    mask = (rng.random((B, n, n)) < np.array([0.2, 0.35, 0.6])[:, None, None]).astype(np.uint8)      # different M0 per problem
    maskT = plan.sel_from_dense(dev(mask))
    bits = plan.pack_mask(maskT)
    # pack_mask against NumPy
    want = np.packbits(np.swapaxes(mask, 1, 2).reshape(B, n, n // 32, 32), axis=-1, bitorder='little').view(np.uint32).reshape(B, n, n // 32)
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), want)
    z = dev(rng.standard_normal((B, n, n)), dt)
    Y = rng.standard_normal((B, n, n)) + 1j * rng.standard_normal((B, n, n))
    YT = torch.from_numpy(np.ascontiguousarray(np.swapaxes(Y, 1, 2))).to('cuda', cdt).contiguous()
    yh = plan.pack_y(YT, maskT)
    g_u8 = plan.grad(z, maskT, yh=yh)
    g_bits = plan.grad(z, bits=bits, yh=yh)
    assert torch.equal(g_u8, g_bits)
    g_yt = plan.grad(z, bits=bits, YT=YT)
    assert (g_yt - g_u8).abs().max().item() <= tol * max(1.0, g_u8.abs().max().item())
    # per-problem scale
    av = dev(np.array([0.5, 2.0, -3.0]), dt)
    g_av = plan.grad(z, bits=bits, yh=yh, alpha=0.25, alpha_vec=av)
    assert (g_av - 0.25 * av[:, None, None] * g_u8).abs().max().item() <= tol * max(1.0, g_u8.abs().max().item())
    # hashed minibatch == its materialised selector, with and without the data term
    mb = 150
    selbits = torch.empty((3, B, n, n // 32), dtype=torch.int32, device='cuda')
    mbd = plan.draw_thresholds(bits, mb, seed=11, step0=5, nsteps=3, selbits=selbits)
    for j in range(3):
        selT = plan.sel_from_thresholds(bits, mbd[j])
        s = selT.cpu().numpy()
        assert (s.reshape(B, -1).sum(1) == mb).all() and (s <= maskT.cpu().numpy()).all()
        assert torch.equal(plan.pack_mask(selT), selbits[j])       # the emitted bits ARE the thresholded selection
        w = dev(rng.standard_normal((B, n, n)), dt)
        a1 = plan.grad(z, selT, b=w, alpha=0.3, beta=1.0, c1=z)
        a2 = plan.grad(z, bits=selbits[j], b=w, alpha=0.3, beta=1.0, c1=z)
        assert torch.equal(a1, a2)
        b1 = plan.grad(z, selT, YT=YT)
        b2 = plan.grad(z, bits=selbits[j], YT=YT)
        assert torch.equal(b1, b2)
        b3 = plan.grad(z, selT, yh=plan.pack_y(YT, selT))
        assert (b1 - b3).abs().max().item() <= tol * max(1.0, b3.abs().max().item())


def test_saga_table_update_and_generic_draws(ops):
    """pnp_saga_table_update == its five NumPy lines (incl. prev aliasing the replaced row); generic draws over M
    measurements: exactly mb members, indicator == ascending row list, deterministic, different per step / problem."""
    rng = np.random.default_rng(3)
    n = 5000
    for dt in (torch.float64, torch.float32):
        z, g, slot, prev, ts = (rng.standard_normal(n) for _ in range(5))
        zd, gd, sd, pd, td = (dev(v, dt) for v in (z, g, slot, prev, ts))
        ops.saga_table_update(zd, gd, sd, pd, td, 0.7, 0.25)
        s2 = ts + g - slot
        tol = 1e-14 if dt == torch.float64 else 2e-6
        np.testing.assert_allclose(zd.cpu().numpy(), z - 0.7 * ((g - prev) + s2 * 0.25), atol=tol * 10)
        np.testing.assert_allclose(td.cpu().numpy(), s2, atol=tol * 10)
        assert torch.equal(sd, gd)
        # prev is the row being replaced
        zd, gd, sd, td = (dev(v, dt) for v in (z, g, slot, ts))
        ops.saga_table_update(zd, gd, sd, sd, td, 0.7, 0.25)
        np.testing.assert_allclose(zd.cpu().numpy(), z - 0.7 * ((g - slot) + s2 * 0.25), atol=tol * 10)
    M, B, mb = 4097, 4, 333
    mbd = ops.draw_thresholds(M, B, mb, seed=5, step0=2, nsteps=2)
    sel = ops.indicator_from_thresholds(M, mbd[0]).cpu().numpy()
    assert (sel.sum(1) == mb).all()
    rows = ops.rows_from_thresholds(M, mb, mbd[0]).cpu().numpy()
    for b in range(B):
        assert np.array_equal(rows[b], np.flatnonzero(sel[b]))
    assert torch.equal(mbd, ops.draw_thresholds(M, B, mb, seed=5, step0=2, nsteps=2))
    sel1 = ops.indicator_from_thresholds(M, mbd[1]).cpu().numpy()
    assert not np.array_equal(sel, sel1) and not np.array_equal(sel[0], sel[1])
    assert torch.equal(mbd[1], ops.draw_thresholds(M, B, mb, seed=5, step0=3, nsteps=1)[0])       # step0 + slot == step
    full = ops.indicator_from_thresholds(M, ops.draw_thresholds(M, B, M, seed=1, step0=0)[0]).cpu().numpy()
    assert full.all()
    idx = torch.from_numpy(rows.astype(np.int32)).cuda()
    assert np.array_equal(ops.indicator_from_indices(idx, M).cpu().numpy(), sel)


@pytest.mark.parametrize('n,dt', [(64, torch.float64), (256, torch.float64), (256, torch.float32), (128, torch.float32)])
def test_small_batch_prox_split_is_bit_identical(ops, n, dt, monkeypatch):
    """Up to 32 images the prox runs as W/16 single-wave workgroups per image in two launches (so that a lone image is
    not confined to one CU); same summation trees as the one-workgroup kernel: bit-identical output, noise estimate and
    error sum, with and without a given sigma, estimate-only mode included."""
    rng = np.random.default_rng(n)
    B = 3
    p = np.pad(rng.random((B, n, n)), ((0, 0), (2, 2), (2, 2)), mode='wrap')
    smooth = sum(p[:, i:i + n, j:j + n] for i in range(5) for j in range(5)) / 25.0
    xrec = dev(smooth, dt)
    z = dev(smooth + 0.05 * rng.standard_normal((B, n, n)), dt)
    sig_in = dev(np.array([0.03, 0.05, 0.0]), dt)
    res = {}
    for split in (True, False):
        if split:
            monkeypatch.delenv('PNP_PROX_NO_SPLIT', raising=False)
        else:
            monkeypatch.setenv('PNP_PROX_NO_SPLIT', '1')
        a = ops.prox_tv(z, xrec=xrec, sigma_modifier=1.2)
        b = ops.prox_tv(z, sigma_in=sig_in, fallback_sigma=0.07, xrec=xrec)
        c = ops.sigma_est(z)
        zz = z.clone()
        d = ops.prox_tv(zz, xrec=xrec, out=zz, sigma_modifier=1.2)          # in place, twice in a row (counter reset)
        d2 = ops.prox_tv(zz, xrec=xrec, out=zz, sigma_modifier=1.2)
        res[split] = [t.clone() for t in (*a, *b, c, d[0], d[1], d2[1])]
    monkeypatch.delenv('PNP_PROX_NO_SPLIT', raising=False)
    for u, v in zip(res[True], res[False]):
        assert torch.equal(u, v)
    assert not torch.equal(res[True][0], z)
