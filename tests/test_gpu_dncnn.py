"""GPU parity of the DnCNN prox (f32 MFMA implicit-GEMM conv stack) against golden vectors from the
reference network class on the reference weights (torch CPU fp32) and against the oracle.
Floating-point kernel: tolerance 2e-5 absolute on O(1) activations / residuals (fp32 summation-order
and BatchNorm-folding differences only)."""
import numpy as np
import pytest
import torch
from conftest import golden

from oracle import denoise as od

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def W15():
    return dict(golden('dncnn_noise15.npz'))


@pytest.fixture(scope='module')
def io():
    return golden('dncnn_io.npz')


def dev(x, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to('cuda', dtype)


@pytest.mark.parametrize('n', [64, 256])
def test_forward_vs_reference_net(W15, io, n):
    from pnp_svrg_amd import ops
    plan = ops.DncnnPlan(W15, n, n, 1)
    r = plan.forward(dev(io[f'net{n}_in'][None])).cpu().numpy()[0]
    ref = io[f'net{n}_out']
    assert np.abs(r - ref).max() <= 2e-5, np.abs(r - ref).max()


def test_forward_batch_and_asymmetry(W15, io):
    """Batch of 3 different images (incl. a transposed one: catches row/col or cin/cout swaps)."""
    from pnp_svrg_amd import ops
    x = io['net64_in']
    xb = np.stack([x, x.T.copy(), np.roll(x, 5, axis=1)])
    plan = ops.DncnnPlan(W15, 64, 64, 3)
    r = plan.forward(dev(xb)).cpu().numpy()
    for i in range(3):
        ref = od.dncnn_forward(W15, xb[i])
        assert np.abs(r[i] - ref).max() <= 2e-5


def test_rect_image(W15):
    """H != W (H multiple of 8, W of 32)."""
    from pnp_svrg_amd import ops
    rng = np.random.default_rng(0)
    x = rng.random((40, 96)).astype(np.float32)
    plan = ops.DncnnPlan(W15, 40, 96, 1)
    r = plan.forward(dev(x[None])).cpu().numpy()[0]
    assert np.abs(r - od.dncnn_forward(W15, x)).max() <= 2e-5


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_denoise_wrapper(W15, io, g_denoise, dtype):
    from pnp_svrg_amd import ops
    for n, zkey, okey in ((64, 's64_z0', 'den64_s15'), (256, 'r256_z0', 'den256_s15')):
        z = g_denoise[zkey]
        plan = ops.DncnnPlan(W15, n, n, 1)
        xrec = np.clip(z, 0, 1)
        out, sse = plan.denoise(dev(z[None], dtype), 15, xrec=dev(xrec[None], dtype))
        o = out.cpu().numpy()[0].astype(np.float64)
        assert np.abs(o - io[okey]).max() <= 3e-5
        assert sse.item() == pytest.approx(((xrec - io[okey]) ** 2).sum(), rel=1e-4)
        # in place
        zt = dev(z[None], dtype)
        plan.denoise(zt, 15, out=zt)
        assert torch.equal(zt, out)


def test_winograd_vs_direct(W15, io):
    """The three fp32 conv kernels (Winograd F(4x4,3x3) default, F(2,3) along x, direct fmaf-chain) agree to fp32
    rounding, and all match the reference network."""
    from pnp_svrg_amd import ops
    x = dev(io['net256_in'][None])
    rw = ops.DncnnPlan(W15, 256, 256, 1, winograd=True).forward(x).cpu().numpy()[0]
    rd = ops.DncnnPlan(W15, 256, 256, 1, winograd=False).forward(x).cpu().numpy()[0]
    r5 = ops.DncnnPlan(W15, 256, 256, 1, winograd=5).forward(x).cpu().numpy()[0]       # F(4x4,3x3): the default
    assert np.array_equal(r5, ops.DncnnPlan(W15, 256, 256, 1).forward(x).cpu().numpy()[0])
    assert not np.array_equal(rw, rd) and not np.array_equal(r5, rw)   # really three different kernels
    assert np.abs(rw - rd).max() <= 1e-5 and np.abs(r5 - rd).max() <= 1e-5
    assert np.abs(rw - io['net256_out']).max() <= 2e-5 and np.abs(rd - io['net256_out']).max() <= 2e-5
    assert np.abs(r5 - io['net256_out']).max() <= 2e-5
    print('max |conv kernel - reference net|: F(4x4,3x3) %.2e  F(2,3) %.2e  direct %.2e' % (
        np.abs(r5 - io['net256_out']).max(), np.abs(rw - io['net256_out']).max(), np.abs(rd - io['net256_out']).max()))
    # several tiles per persistent workgroup in the XCD-aware order (tilewalk.h), and a count that does not divide
    # (plain walk): every image of a batch must equal its single-image result, for all conv kernels
    rng = np.random.default_rng(5)
    for B in (6, 5):
        xb = rng.random((B, 256, 256)).astype(np.float32)
        xb[B - 1] = io['net256_in']
        for mode in (5, 1, 0):
            rb = ops.DncnnPlan(W15, 256, 256, B, winograd=mode).forward(dev(xb)).cpu().numpy()
            one = ops.DncnnPlan(W15, 256, 256, 1, winograd=mode).forward(dev(xb[1:2])).cpu().numpy()[0]
            assert np.array_equal(rb[1], one), (B, mode)
            assert np.abs(rb[B - 1] - io['net256_out']).max() <= 2e-5


def test_wino44_shapes_and_edges():
    """The F(4x4,3x3) kernel on non-square images whose 8 x 64 regions all touch an edge, batches whose region count does
    not divide the grid (plain tile walk) and single-region images: equal to the direct kernel to fp32 rounding; constant
    offset far from zero (every halo value matters); a width outside its grid falls back to F(2,3)."""
    from pnp_svrg_amd import ops
    from pnp_svrg_amd.denoisers import random_dncnn_weights
    w = random_dncnn_weights(5, seed=3)
    rng = np.random.default_rng(8)
    for (H, Wd, B) in ((8, 64, 1), (72, 128, 3), (128, 64, 5), (64, 192, 2)):
        x = rng.random((B, H, Wd)).astype(np.float32) + 3.0
        r0 = ops.DncnnPlan(w, H, Wd, B, winograd=0).forward(dev(x)).cpu().numpy()
        r5 = ops.DncnnPlan(w, H, Wd, B, winograd=5).forward(dev(x)).cpu().numpy()
        assert np.abs(r5 - r0).max() <= 2e-5 * max(1.0, np.abs(r0).max()), (H, Wd, B, np.abs(r5 - r0).max())
    x96 = rng.random((1, 40, 96)).astype(np.float32)                    # W % 64 != 0: the default is F(2,3) there
    rd = ops.DncnnPlan(w, 40, 96, 1).forward(dev(x96)).cpu().numpy()
    assert np.array_equal(rd, ops.DncnnPlan(w, 40, 96, 1, winograd=1).forward(dev(x96)).cpu().numpy())
    with pytest.raises(Exception):
        ops.DncnnPlan(w, 40, 96, 1, winograd=5)


def test_wino44_run_to_run_identical(W15):
    """Repeated forward passes through the F(4x4,3x3) layers are bit-identical, with HBM traffic from a second stream under
    half of them (a stale accumulator copy, a missed DMA wait or an LDS race would differ from run to run); both region
    forms (1 image: 4 x 64 regions, 7 images: 8 x 64) and region counts that do not fill the last wave of workgroups."""
    from pnp_svrg_amd import ops
    side = torch.cuda.Stream()
    junk = torch.empty(32 * 1024 * 1024, device='cuda')
    for B in (1, 7):
        plan = ops.DncnnPlan(W15, 256, 256, B, winograd=5)
        x = torch.rand(B, 256, 256, device='cuda')
        first = plan.forward(x).clone()
        for i in range(12):
            if i % 2:
                with torch.cuda.stream(side):
                    junk.mul_(1.0001)
            assert torch.equal(plan.forward(x), first), (B, i)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['SimpleCNN', 'RealSN_SimpleCNN'])
def test_simplecnn_family(name):
    """the 4-layer SimpleCNN / RealSN_SimpleCNN denoisers run through the same plan (n_mid = 2,
    no BatchNorm).  Golden output = the reference's own class (incl. its spectral-norm eval hook for the RealSN
    checkpoint) on its own weights."""
    from pnp_svrg_amd import ops
    g = golden('simplecnn_noise15.npz')
    w = {'n_layers': np.int64(4)}
    for i in range(4):
        w[f'conv{i}.weight'] = g[f'{name}_conv{i}.weight']
    for wino in (5, 1, 0):
        r = ops.DncnnPlan(w, 64, 64, 1, winograd=wino).forward(dev(g['net64_in'][None])).cpu().numpy()[0]
        assert np.abs(r - g[f'{name}_out']).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('winograd', [True, False, 5, 6])
def test_mmo_denoiser_vs_reference(winograd):
    """MMODenoiser (20-layer bias / LeakyReLU(0.01) / skip net, transposed input, both clamps;
    reference denoisers/MMODenoise.py:18-40,73-128) through the MFMA conv stack vs the reference's own classes
    (tests/golden/make_golden_mmo.py), NumPy protocol: float32 result, t advances."""
    from pnp_svrg_amd.denoisers import MMODenoiser, mmo_weights_from_state_dict
    import os
    g = golden('mmo_seeded.npz')
    names = ['in_conv'] + [f'conv_list.{i}' for i in range(18)] + ['out_conv']
    sd = {}
    for i, n in enumerate(names):
        sd['module.' + n + '.weight'] = torch.from_numpy(g[f'conv{i}.weight'])
        sd['module.' + n + '.bias'] = torch.from_numpy(g[f'conv{i}.bias'])
    old = os.environ.get('PNP_DNCNN_WINOGRAD')
    os.environ['PNP_DNCNN_WINOGRAD'] = str(int(winograd))     # 6 / 5 / 1 = Winograd kernels (LeakyReLU builds), 0 = direct
    try:
        den = MMODenoiser(model=sd, channels=1)
        for name in ('sq', 'rect'):
            y = den.denoise(g[f'{name}_in'])
            assert y.dtype == np.float32 and y.shape == g[f'{name}_out'].shape
            assert np.abs(y - g[f'{name}_out']).max() <= 2e-5, (name, np.abs(y - g[f'{name}_out']).max())
            assert y.min() == 0.0 and y.max() == 1.0
    finally:
        if old is None:
            os.environ.pop('PNP_DNCNN_WINOGRAD')
        else:
            os.environ['PNP_DNCNN_WINOGRAD'] = old
    assert den.t == 2
    w = mmo_weights_from_state_dict(sd)
    assert int(w['n_layers']) == 20 and w['transpose_taps'] is True
    with pytest.raises(NotImplementedError):
        MMODenoiser(model=sd, channels=3)


@pytest.mark.gpu
def test_mmo_device_batch_and_sse():
    """denoise_device on a batch (f64 storage): every image equals the single-image result; the fused squared error
    equals the one recomputed from the output."""
    from pnp_svrg_amd.denoisers import MMODenoiser
    g = golden('mmo_seeded.npz')
    w = {k: g[k] for k in g.files if k.startswith('conv') or k in ('n_layers', 'negative_slope')}
    den = MMODenoiser(weights=w, channels=1)
    x = g['rect_in']
    z = dev(np.stack([x, x[::-1].copy(), 0.5 * x]), torch.float64)
    xrec = dev(np.stack([g['rect_out']] * 3).astype(np.float64), torch.float64)
    out, sse, _ = den.denoise_device(z, xrec=xrec)
    o = out.cpu().numpy()
    assert np.abs(o[0] - g['rect_out']).max() <= 2e-5
    for b in range(3):
        single = den.denoise(z[b].cpu().numpy())
        assert np.abs(single - o[b]).max() <= 1e-6
    ref = ((xrec.cpu().numpy() - o) ** 2).sum(axis=(1, 2))
    assert np.allclose(sse.cpu().numpy(), ref, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('scale', [1.0, 1e-3, 50.0])
def test_conv_kernels_against_float64(scale):
    """Error of each conv kernel against a float64 evaluation of the same 3-layer net (1->64, 64->64 + bias + ReLU,
    64->1; random weights), at activation scales from 1e-3 to 50."""
    import torch.nn.functional as F
    from pnp_svrg_amd import ops
    rng = np.random.default_rng(11)
    n = 64
    w = {'n_layers': np.int64(3),
         'conv0.weight': (rng.standard_normal((64, 1, 3, 3)) * scale).astype(np.float32),
         'conv1.weight': (rng.standard_normal((64, 64, 3, 3)) / 24.0).astype(np.float32),
         'conv1.bias': (rng.standard_normal(64) * 0.1 * scale).astype(np.float32),
         'conv2.weight': (rng.standard_normal((1, 64, 3, 3)) / 24.0).astype(np.float32)}
    x = rng.random((2, n, n)).astype(np.float32)
    t = torch.from_numpy(x).double()[:, None]
    t = F.relu(F.conv2d(t, torch.from_numpy(w['conv0.weight']).double(), padding=1))
    t = F.relu(F.conv2d(t, torch.from_numpy(w['conv1.weight']).double(), torch.from_numpy(w['conv1.bias']).double(), padding=1))
    ref = F.conv2d(t, torch.from_numpy(w['conv2.weight']).double(), padding=1)[:, 0].numpy()
    err = {}
    for mode in (0, 1, 5):
        r = ops.DncnnPlan(w, n, n, 2, winograd=mode).forward(dev(x)).cpu().numpy().astype(np.float64)
        err[mode] = np.abs(r - ref).max() / np.abs(ref).max()
    print(f'scale {scale}: relative max error vs float64 -- fp32 direct {err[0]:.2e}, F(2,3) {err[1]:.2e}, F(4x4,3x3) {err[5]:.2e}')
    # the two-dimensional transform pays for its 4x fewer multiply-adds with ~4-5x the rounding error of the direct form on
    # white-noise weights (transform entries up to 8 and 1/24); on the reference's weights: 8e-7 vs 4e-7 (test_winograd_vs_direct)
    assert err[0] < 2e-6 and err[1] < 2e-6 and err[5] < 1e-5


def test_wino44_guard_bands():
    """The F(4x4,3x3) layer (`k_mid_wino44`: LDS-DMA halo loads that lean on the buffer descriptor's range check, a weight
    stream prefetched a ring ahead, a prefetch of the NEXT region's first chunks) on CALLER-provided buffers with canary-filled
    guard bands in front of and behind the input, the output and the weights: both region forms (8 x 64: `Geo<2>`, 4 x 64:
    `Geo<1>`), the second-launch tail path (a last, at most half-full wave goes through the 4 x 64 form), a 72 x 128 and a
    256 x 256 B = 5 case.  Asserts (a) every canary intact -- nothing is written outside `out`; (b) the result equals the direct
    kernel's on the same input to fp32 rounding -- a read that strayed into a band (NaN canaries) would poison it."""
    from pnp_svrg_amd import ops
    from pnp_svrg_amd.denoisers import random_dncnn_weights
    w = random_dncnn_weights(4, seed=9)
    rng = np.random.default_rng(21)
    GUARD = 1 << 18                                              # floats (1 MiB) on each side
    # rows: 0 = the production choice (256 x 256 B = 5: 640 regions of 8 x 64 = two full waves of 256 + a 128-region rest -> the
    # second launch in the 4 x 64 form), 1 / 2 = one form for the whole layer
    cases = [(72, 128, 3, 0), (256, 256, 5, 0), (256, 256, 5, 1), (256, 256, 5, 2), (8, 64, 1, 0), (64, 192, 2, 0), (72, 128, 3, 2)]
    for (H, Wd, B, rows) in cases:
        n = B * 64 * H * Wd
        plan5 = ops.DncnnPlan(w, H, Wd, B, winograd=5)
        plan0 = ops.DncnnPlan(w, H, Wd, B, winograd=0)
        x = rng.standard_normal(n).astype(np.float32)

        def banded(nfloats, fill):
            t = torch.full((nfloats + 2 * GUARD,), float('nan'), dtype=torch.float32, device='cuda')
            t[GUARD:GUARD + nfloats] = fill
            return t
        xin = banded(n, torch.from_numpy(x).cuda())
        yout = banded(n, 0.0)
        wk = plan5.debug_w44_weights(1)
        wb = banded(wk.numel(), wk)
        vin, vout, vw = xin[GUARD:GUARD + n].view(B, 64, H, Wd), yout[GUARD:GUARD + n].view(B, 64, H, Wd), wb[GUARD:GUARD + wk.numel()]
        assert vin.data_ptr() % 16 == 0 and vout.data_ptr() % 16 == 0 and vw.data_ptr() % 16 == 0
        plan5.debug_mid_layer(1, vin, vout, w44=vw, rows=rows)
        ref = torch.empty((B, 64, H, Wd), dtype=torch.float32, device='cuda')
        plan0.debug_mid_layer(1, torch.from_numpy(x).cuda().view(B, 64, H, Wd), ref)
        torch.cuda.synchronize()
        for t, nn in ((xin, n), (yout, n), (wb, wk.numel())):
            assert torch.isnan(t[:GUARD]).all() and torch.isnan(t[GUARD + nn:]).all(), (H, Wd, B, rows)
        assert torch.equal(xin[GUARD:GUARD + n], torch.from_numpy(x).cuda()) and torch.equal(vw, wk)
        assert torch.isfinite(vout).all()
        assert (vout - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), (H, Wd, B, rows)


def test_bf16x3_conv_mode(W15, io):
    """Conv mode 6 (`k_mid_wino44b`: F(4x4,3x3) on the bf16 matrix cores, every fp32 factor split exactly into three bf16 terms, six
    products) is fp32-class: (a) the reference network's own output within the bound of the fp32 kernels; (b) against float64 at
    three activation scales no worse than the fp32 F(4x4,3x3) kernel (bf16 keeps fp32's exponent range: nothing depends on the
    scale); (c) shapes whose regions all touch an edge, region counts that do not divide the grid, images of a batch equal to
    their single-image results; (d) run-to-run identical; (e) NaN guard bands around input and output intact."""
    import torch.nn.functional as F
    from pnp_svrg_amd import ops
    from pnp_svrg_amd.denoisers import random_dncnn_weights
    x = dev(io['net256_in'][None])
    r6 = ops.DncnnPlan(W15, 256, 256, 1, winograd=6).forward(x).cpu().numpy()[0]
    r5 = ops.DncnnPlan(W15, 256, 256, 1, winograd=5).forward(x).cpu().numpy()[0]
    assert not np.array_equal(r6, r5)                                       # really another kernel
    assert np.abs(r6 - io['net256_out']).max() <= 2e-5 and np.abs(r6 - r5).max() <= 1e-5
    print('max |bf16x3 F(4x4,3x3) - reference net| %.2e   (fp32 F(4x4,3x3): %.2e)' % (
        np.abs(r6 - io['net256_out']).max(), np.abs(r5 - io['net256_out']).max()))
    # (b)
    for scale in (1.0, 1e-3, 50.0):
        rng = np.random.default_rng(11)
        n = 64
        w = {'n_layers': np.int64(3),
             'conv0.weight': (rng.standard_normal((64, 1, 3, 3)) * scale).astype(np.float32),
             'conv1.weight': (rng.standard_normal((64, 64, 3, 3)) / 24.0).astype(np.float32),
             'conv1.bias': (rng.standard_normal(64) * 0.1 * scale).astype(np.float32),
             'conv2.weight': (rng.standard_normal((1, 64, 3, 3)) / 24.0).astype(np.float32)}
        xs = rng.random((2, n, n)).astype(np.float32)
        t = torch.from_numpy(xs).double()[:, None]
        t = F.relu(F.conv2d(t, torch.from_numpy(w['conv0.weight']).double(), padding=1))
        t = F.relu(F.conv2d(t, torch.from_numpy(w['conv1.weight']).double(), torch.from_numpy(w['conv1.bias']).double(), padding=1))
        ref = F.conv2d(t, torch.from_numpy(w['conv2.weight']).double(), padding=1)[:, 0].numpy()
        err = {m: np.abs(ops.DncnnPlan(w, n, n, 2, winograd=m).forward(dev(xs)).cpu().numpy().astype(np.float64) - ref).max() / np.abs(ref).max()
               for m in (5, 6)}
        print(f'scale {scale}: relative max error vs float64 -- fp32 F(4x4,3x3) {err[5]:.2e}, bf16x3 F(4x4,3x3) {err[6]:.2e}')
        assert err[6] < 1e-5 and err[6] <= 1.25 * err[5]
    # (c)
    w = random_dncnn_weights(5, seed=3)
    rng = np.random.default_rng(8)
    for (H, Wd, B) in ((8, 64, 1), (72, 128, 3), (128, 64, 5), (64, 192, 2)):
        xs = rng.random((B, H, Wd)).astype(np.float32) + 3.0
        r0 = ops.DncnnPlan(w, H, Wd, B, winograd=0).forward(dev(xs)).cpu().numpy()
        rb = ops.DncnnPlan(w, H, Wd, B, winograd=6).forward(dev(xs)).cpu().numpy()
        assert np.abs(rb - r0).max() <= 2e-5 * max(1.0, np.abs(r0).max()), (H, Wd, B, np.abs(rb - r0).max())
    with pytest.raises(Exception):
        ops.DncnnPlan(w, 40, 96, 1, winograd=6)
    for B in (6, 5):
        xb = rng.random((B, 256, 256)).astype(np.float32)
        rb = ops.DncnnPlan(W15, 256, 256, B, winograd=6).forward(dev(xb)).cpu().numpy()
        one = ops.DncnnPlan(W15, 256, 256, 1, winograd=6).forward(dev(xb[1:2])).cpu().numpy()[0]
        assert np.array_equal(rb[1], one), B
    # (d)
    plan = ops.DncnnPlan(W15, 256, 256, 7, winograd=6)
    xr = torch.rand(7, 256, 256, device='cuda')
    first = plan.forward(xr).clone()
    for _ in range(5):
        assert torch.equal(plan.forward(xr), first)
    # (e)
    w4 = random_dncnn_weights(4, seed=9)
    GUARD = 1 << 18
    for (H, Wd, B) in ((72, 128, 3), (256, 256, 5), (8, 64, 1)):
        n = B * 64 * H * Wd
        plan6, plan0 = ops.DncnnPlan(w4, H, Wd, B, winograd=6), ops.DncnnPlan(w4, H, Wd, B, winograd=0)
        xs = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda()
        xin = torch.full((n + 2 * GUARD,), float('nan'), dtype=torch.float32, device='cuda')
        yout = torch.full((n + 2 * GUARD,), float('nan'), dtype=torch.float32, device='cuda')
        xin[GUARD:GUARD + n] = xs
        yout[GUARD:GUARD + n] = 0.0
        vin, vout = xin[GUARD:GUARD + n].view(B, 64, H, Wd), yout[GUARD:GUARD + n].view(B, 64, H, Wd)
        plan6.debug_mid_layer(1, vin, vout)
        ref = torch.empty((B, 64, H, Wd), dtype=torch.float32, device='cuda')
        plan0.debug_mid_layer(1, xs.view(B, 64, H, Wd), ref)
        torch.cuda.synchronize()
        for t in (xin, yout):
            assert torch.isnan(t[:GUARD]).all() and torch.isnan(t[GUARD + n:]).all(), (H, Wd, B)
        assert torch.equal(xin[GUARD:GUARD + n], xs) and torch.isfinite(vout).all()
        assert (vout - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()), (H, Wd, B)
