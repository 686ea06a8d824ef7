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
    """The two conv kernels (Winograd F(2,3) default, direct fmaf-chain) agree to fp32 rounding, and both
    match the reference network."""
    from pnp_svrg_amd import ops
    x = dev(io['net256_in'][None])
    rw = ops.DncnnPlan(W15, 256, 256, 1, winograd=True).forward(x).cpu().numpy()[0]
    rd = ops.DncnnPlan(W15, 256, 256, 1, winograd=False).forward(x).cpu().numpy()[0]
    assert not np.array_equal(rw, rd)                          # really two different kernels
    assert np.abs(rw - rd).max() <= 1e-5
    assert np.abs(rw - io['net256_out']).max() <= 2e-5 and np.abs(rd - io['net256_out']).max() <= 2e-5


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
    for wino in (1, 0, 2):
        r = ops.DncnnPlan(w, 64, 64, 1, winograd=wino).forward(dev(g['net64_in'][None])).cpu().numpy()[0]
        assert np.abs(r - g[f'{name}_out']).max() <= 2e-5
