"""GPU parity of the five PnP loops (drop-in `algorithms` / `problems` / `denoisers` packages)
against golden traces recorded from the real reference (tests/golden/traces*.npz) and against
the oracle run on the same seeds.

Tolerances: f64 device path -> identical rounded-PSNR traces, |z - z_ref| <= 1e-9.
            f32 device path (production dtype) -> every PSNR entry within +-0.01 dB (the
            north-star tolerance; PSNR is rounded to 0.01 dB by the reference), |z - z_ref| <= 5e-4.
"""
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN

from oracle import loops as ol

pytestmark = pytest.mark.gpu
IMG256 = os.path.join(GOLDEN, 'synth256.png')
IMG64 = os.path.join(GOLDEN, 'synth64.png')


@pytest.fixture(scope='module')
def api():
    import algorithms
    import problems
    import denoisers
    return algorithms, problems, denoisers


def _csmri(problems, path, n, dtype):
    np.random.seed(0)
    return problems.CSMRI(path, H=n, W=n, sample_prob=0.2, snr=20., dtype=dtype)


def _check(r, g, name, dtype):
    ps, ref = np.array(r['psnr_per_iter']), g[f'{name}_psnr']
    assert len(ps) == len(ref)
    if dtype == torch.float64:
        assert list(ps) == list(ref)
        np.testing.assert_allclose(r['z'], g[f'{name}_z'], rtol=0, atol=1e-9)
    else:
        assert np.abs(ps - ref).max() <= 0.01 + 1e-9
        np.testing.assert_allclose(r['z'], g[f'{name}_z'], rtol=0, atol=5e-4)


RUNS = {
    'gd': lambda A, p, d, c: A.pnp_gd(p, d, 5e2, 61, verbose=False, converge_check=False, clock=c),
    'sgd': lambda A, p, d, c: A.pnp_sgd(p, d, 5e2, 51, 200, verbose=False, converge_check=False, lr_decay=0.95, clock=c),
    'svrg': lambda A, p, d, c: A.pnp_svrg(p, d, 5e2, 60, 4, 200, verbose=False, converge_check=False, clock=c),
    'saga': lambda A, p, d, c: A.pnp_saga(p, d, 5e2, 53, 200, hist_size=5, verbose=False, converge_check=False, clock=c),
    'sarah': lambda A, p, d, c: A.pnp_sarah(p, d, 5e2, 70, 4, 200, verbose=False, converge_check=False, lr_decay=0.9, clock=c),
    'gd_conv': lambda A, p, d, c: A.pnp_gd(p, d, 5e2, 2000, verbose=False, converge_check=True, clock=c),
    'svrg_conv': lambda A, p, d, c: A.pnp_svrg(p, d, 5e2, 5000, 4, 200, verbose=False, converge_check=True,
                                               diverge_check=True, clock=c),
}


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('name', list(RUNS))
def test_traces64(api, g_traces64, name, dtype):
    A, P, D = api
    p = _csmri(P, IMG64, 64, dtype)
    np.random.seed(1)
    r = RUNS[name](A, p, D.TVDenoiser(), ol.CountingClock())
    if dtype == torch.float32 and name.endswith('_conv'):
        # the stopping rule compares two 0.01-dB-rounded PSNRs: an f32 run may stop an iteration
        # earlier/later; check the common prefix and the final quality instead
        ps, ref = np.array(r['psnr_per_iter']), g_traces64[f'{name}_psnr']
        m = min(len(ps), len(ref))
        assert abs(len(ps) - len(ref)) <= 2 and np.abs(ps[:m] - ref[:m]).max() <= 0.01 + 1e-9
        return
    _check(r, g_traces64, name, dtype)
    assert r['algo_name'] == {'gd': 'PnP GD', 'sgd': 'PnP SGD', 'svrg': 'PnP SVRG', 'saga': 'pnp_saga',
                              'sarah': 'pnp_sarah', 'gd_conv': 'PnP GD', 'svrg_conv': 'PnP SVRG'}[name]
    assert list(r['time_per_iter']) == list(g_traces64[f'{name}_time'])
    assert [r['gradient_time'], r['denoise_time']] == list(g_traces64[f'{name}_gt_dt'])
    assert isinstance(r['z'], np.ndarray) and r['z'].dtype == np.float64 and r['z'].shape == (p.N,)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_true_svrg_64(api, g_traces64, dtype):
    A, P, D = api
    p = _csmri(P, IMG64, 64, dtype)
    np.random.seed(1)
    r = A.pnp_svrg(p, D.TVDenoiser(), 5e2, 2 + 3 * 23 - 1, 4, 200, verbose=False, converge_check=False,
                   clock=ol.CountingClock(), variant='svrg')
    _check(r, g_traces64, 'truesvrg', dtype)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
@pytest.mark.parametrize('variant,key', [('reference', 'svrg'), ('svrg', 'truesvrg')])
def test_traces256(api, g_traces256, variant, key, dtype):
    """BASELINE config 2 shape: 256x256 CSMRI, 20 % mask, TV prox, pnp_svrg, 4 outer x 10 inner."""
    A, P, D = api
    p = _csmri(P, IMG256, 256, dtype)
    np.random.seed(1)
    r = A.pnp_svrg(p, D.TVDenoiser(), 2e3, 2 + 4 * 53 - (1 if variant == 'svrg' else 0), 10, 1000, verbose=False,
                   converge_check=False, clock=ol.CountingClock(), variant=variant)
    _check(r, g_traces256, key, dtype)


def test_problem_surface(api, g_csmri):
    """Attributes / conventions callers read."""
    A, P, D = api
    g = g_csmri
    p = _csmri(P, IMG64, 64, torch.float64)
    for a in ('H', 'W', 'N', 'M', 'M0', 'X', 'Xrec', 'Xinit', 'Y', 'Y0', 'mask', 'sigma', 'snr', 'pname', 'lrH', 'lrW'):
        assert hasattr(p, a)
    assert p.get_item('M0') == int(g['s64_M0']) and p.pname == 'csmri'
    assert np.array_equal(p.mask, g['s64_mask'])
    np.testing.assert_allclose(p.Xinit, g['s64_Xinit'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(p.Y, g['s64_Y'], rtol=0, atol=1e-9)
    np.random.seed(7)
    mb = p.select_mb(200)
    assert mb.shape == (64, 64) and mb.dtype.kind == 'i' and np.array_equal(mb, g['s64_mb'])
    gf = p.grad_full(p.Xinit)
    assert isinstance(gf, np.ndarray) and gf.shape == (p.N,) and gf.dtype == np.float64
    np.testing.assert_allclose(gf, g['s64_grad_full'], rtol=0, atol=1e-13)
    np.testing.assert_allclose(p.grad_stoch(p.Xinit, mb), g['s64_grad_stoch'], rtol=0, atol=1e-10)
    assert p.PSNR(p.Xinit) == float(g['s64_psnr_init'])
    np.testing.assert_allclose(p.f(p.Xinit), float(g['s64_f']), rtol=1e-9)
    xi = p.Xinit.copy()
    A.pnp_gd(p, D.TVDenoiser(), 5e2, 20, verbose=False, clock=ol.CountingClock())
    assert np.array_equal(p.Xinit, xi)                       # Xinit must not be mutated
    with pytest.raises(Exception):
        P.CSMRI(None, 64, 64)
    with pytest.raises(Exception):
        P.CSMRI(IMG64, 64, 64, snr=20., sigma=0.1)
    with pytest.raises(ValueError):                          # oversize minibatch: prints, then NumPy raises
        p.select_mb(p.M0 + 1)
    with pytest.raises(NotImplementedError):
        P.Problem(IMG64, 64, 64).grad_full(p.Xinit)
    with pytest.raises(NotImplementedError):
        D.Denoise().denoise(None)


def test_tv_denoiser_surface(api, g_denoise):
    A, P, D = api
    g = g_denoise
    d = D.TVDenoiser(dtype=torch.float64)
    out = d.denoise(noisy=g['s64_z0'], sigma_est=float(g['s64_sigma_est']))
    assert d.t == 1 and isinstance(out, np.ndarray) and out.shape == (64, 64)
    np.testing.assert_allclose(out, g['s64_tv'], rtol=0, atol=1e-12)
    d = D.TVDenoiser(denoise_strength=0.07, decay=0.9, dtype=torch.float64)
    np.testing.assert_allclose(d.denoise(noisy=g['s64_z0'], sigma_est=0), g['s64_tv_strength'], rtol=0, atol=1e-12)


def test_foreign_denoiser_and_problem(api, g_traces64):
    """Third-party NumPy-protocol denoisers/problems still plug in (the oracle's classes play that role)."""
    A, P, D = api
    from oracle import denoise as od, problems as op
    p = _csmri(P, IMG64, 64, torch.float64)
    np.random.seed(1)
    r = RUNS['svrg'](A, p, od.TVDenoiser(), ol.CountingClock())          # native problem, foreign denoiser
    assert list(r['psnr_per_iter']) == list(g_traces64['svrg_psnr'])
    np.random.seed(0)
    po = op.CSMRI(IMG64, H=64, W=64, sample_prob=0.2, snr=20.)
    np.random.seed(1)
    r = RUNS['sgd'](A, po, D.TVDenoiser(dtype=torch.float64), ol.CountingClock())   # foreign problem, native denoiser
    assert list(r['psnr_per_iter']) == list(g_traces64['sgd_psnr'])
    np.testing.assert_allclose(r['z'], g_traces64['sgd_z'], rtol=0, atol=1e-9)


def test_tune_wrappers(api):
    A, P, D = api
    p = _csmri(P, IMG64, 64, torch.float32)
    np.random.seed(1)
    r = A.tune_pnp_svrg((5e2, 200, 3, 0.1), p, D.TVDenoiser(), tt=0.2)
    assert set(r) == {'loss', 'status', 'algo_name', 'z', 'time_per_iter', 'psnr_per_iter', 'gradient_time', 'denoise_time'}
    assert r['status'] == 'ok' and r['loss'] == p.PSNR(p.Xinit) - p.PSNR(r['z'])
    r = A.tune_pnp_saga((5e2, 200, 0.1, 4), p, D.TVDenoiser(), tt=0.1)
    assert r['algo_name'] == 'pnp_saga'


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_svrg_dncnn_vs_oracle(api, dtype):
    """BASELINE config 3 shape (CSMRI + DnCNN prox, pnp_svrg), 64x64, reference DnCNN weights:
    the MFMA net inside the loop against the oracle loop with the torch-CPU fp32 net."""
    from conftest import golden
    from oracle import denoise as od, problems as op
    A, P, D = api
    wts = dict(golden('dncnn_noise15.npz'))
    for variant in ('reference', 'svrg'):
        p = _csmri(P, IMG64, 64, dtype)
        np.random.seed(1)
        r = A.pnp_svrg(p, D.RealSN_DnCNNDenoiser('DnCNN', 15, weights=wts), 5e2, 2 + 2 * (3 + 5 * 3), 3, 200,
                       verbose=False, converge_check=False, clock=ol.CountingClock(), variant=variant)
        np.random.seed(0)
        po = op.CSMRI(IMG64, H=64, W=64, sample_prob=0.2, snr=20.)
        np.random.seed(1)
        ro = ol.pnp_svrg(po, od.DnCNNDenoiser(wts, 15), 5e2, 2 + 2 * (3 + 5 * 3), 3, 200, converge_check=False,
                         clock=ol.CountingClock(), variant=variant)
        ps, pso = np.array(r['psnr_per_iter']), np.array(ro['psnr_per_iter'])
        assert len(ps) == len(pso) and np.abs(ps - pso).max() <= 0.01 + 1e-9
        np.testing.assert_allclose(r['z'], ro['z'], rtol=0, atol=2e-4)


def test_dncnn_denoiser_surface(api, g_denoise):
    from conftest import golden
    A, P, D = api
    io = golden('dncnn_io.npz')
    d = D.RealSN_DnCNNDenoiser('DnCNN', 15, weights=dict(golden('dncnn_noise15.npz')))
    out = d.denoise(noisy=g_denoise['s64_z0'], sigma_est=123.0)       # sigma_est is ignored (F12)
    assert d.t == 0 and out.shape == (64, 64) and out.dtype == np.float64
    np.testing.assert_allclose(out, io['den64_s15'], rtol=0, atol=3e-5)
    with pytest.raises(FileNotFoundError):                            # CWD-relative checkpoint path, as the reference
        D.RealSN_DnCNNDenoiser('RealSN_DnCNN', 5)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_csmri_128(api, dtype):
    """128 x 128 (the size the reference's CSMRI/PR notebooks use): rectangular 8 x 16 FFT split.  No golden
    file at this size -> gradients and a short SVRG run against the (golden-pinned) oracle on the same seeds."""
    from oracle import denoise as od, problems as op
    A, P, D = api
    np.random.seed(0)
    p = P.CSMRI(IMG256, H=128, W=128, sample_prob=0.5, snr=10., dtype=dtype)       # PIL resizes 256 -> 128
    np.random.seed(0)
    po = op.CSMRI(IMG256, H=128, W=128, sample_prob=0.5, snr=10.)
    assert np.array_equal(p.mask, po.mask) and p.M0 == po.M0
    np.random.seed(7)
    mb = p.select_mb(500)
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    gf, gfo = p.grad_full(po.Xinit), po.grad_full(po.Xinit)
    assert np.abs(gf - gfo).max() <= tol * np.abs(gfo).max() * 10
    gs, gso = p.grad_stoch(po.Xinit, mb), po.grad_stoch(po.Xinit, mb)
    assert np.abs(gs - gso).max() <= tol * np.abs(gso).max() * 10
    for variant in ('reference', 'svrg'):
        np.random.seed(1)
        r = A.pnp_svrg(p, D.TVDenoiser(), 0.1, 2 + 2 * (3 + 5 * 4), 4, 500, verbose=False, converge_check=False,
                       clock=ol.CountingClock(), variant=variant)
        np.random.seed(1)
        ro = ol.pnp_svrg(po, od.TVDenoiser(), 0.1, 2 + 2 * (3 + 5 * 4), 4, 500, converge_check=False,
                         clock=ol.CountingClock(), variant=variant)
        ps, pso = np.array(r['psnr_per_iter']), np.array(ro['psnr_per_iter'])
        assert len(ps) == len(pso)
        if dtype == torch.float64:
            assert list(ps) == list(pso)
            np.testing.assert_allclose(r['z'], ro['z'], rtol=0, atol=1e-9)
        else:
            assert np.abs(ps - pso).max() <= 0.01 + 1e-9


def test_problem_display_sets_attrs(api, tmp_path):
    """problems/problem.py:64-108: callers (Utilities.display_results) read color_map / prob_dir afterwards."""
    A, P, D = api
    p = _csmri(P, IMG64, 64, torch.float32)
    p.display(color_map='gray', show_measurements=True, save_results=True, save_dir=str(tmp_path) + '/')
    assert p.color_map == 'gray' and p.prob_dir.startswith(str(tmp_path)) and os.path.isdir(p.prob_dir)
    assert sorted(os.listdir(p.prob_dir)) == ['initialization.eps', 'measurements.eps', 'original.eps']


def test_config3_full_size_vs_oracle(api, g_csmri):
    """BASELINE config 3 at full size: 256 x 256 CSMRI (20 % mask) on the photograph fixture, DnCNN prox with the
    reference's sigma=15 weights, pnp_svrg (both directions), f32 device path with the default (Winograd) conv
    kernel: every logged PSNR within +-0.01 dB of the oracle loop (torch-CPU fp32 net) on identical seeds."""
    from conftest import golden
    from oracle import denoise as od, problems as op
    A, P, D = api
    wts = dict(golden('dncnn_noise15.npz'))
    img = g_csmri['r256_img']
    for variant in ('reference', 'svrg'):
        np.random.seed(0)
        p = P.CSMRI(None, H=256, W=256, sample_prob=0.2, snr=20., img=img, dtype=torch.float32)
        assert np.array_equal(p.mask, g_csmri['r256_mask'])
        np.random.seed(1)
        r = A.pnp_svrg(p, D.RealSN_DnCNNDenoiser('DnCNN', 15, weights=wts), 2e3, 2 + 2 * (3 + 5 * 5), 5, 1000,
                       verbose=False, converge_check=False, clock=ol.CountingClock(), variant=variant)
        np.random.seed(0)
        po = op.CSMRI(None, H=256, W=256, sample_prob=0.2, snr=20., img=img)
        np.random.seed(1)
        ro = ol.pnp_svrg(po, od.DnCNNDenoiser(wts, 15), 2e3, 2 + 2 * (3 + 5 * 5), 5, 1000, converge_check=False,
                         clock=ol.CountingClock(), variant=variant)
        ps, pso = np.array(r['psnr_per_iter']), np.array(ro['psnr_per_iter'])
        assert len(ps) == len(pso) == 13
        assert np.abs(ps - pso).max() <= 0.01 + 1e-9, (ps, pso)
        assert ps[-1] > ps[0] + 1.0                              # and it actually reconstructs
        assert np.abs(r['z'] - ro['z']).max() < 5e-4


def test_config3_bf16x3_conv_vs_oracle(api, g_csmri, monkeypatch):
    """The config-3 loop of test_config3_full_size_vs_oracle with the opt-in conv mode 6 (F(4x4,3x3) on three-way bf16 splits):
    every logged PSNR within +-0.01 dB of the oracle loop (torch-CPU fp32 net) on identical seeds."""
    from conftest import golden
    from oracle import denoise as od, problems as op
    A, P, D = api
    monkeypatch.setenv('PNP_DNCNN_WINOGRAD', '6')
    wts = dict(golden('dncnn_noise15.npz'))
    img = g_csmri['r256_img']
    np.random.seed(0)
    p = P.CSMRI(None, H=256, W=256, sample_prob=0.2, snr=20., img=img, dtype=torch.float32)
    np.random.seed(1)
    r = A.pnp_svrg(p, D.RealSN_DnCNNDenoiser('DnCNN', 15, weights=wts), 2e3, 2 + 2 * (3 + 5 * 5), 5, 1000,
                   verbose=False, converge_check=False, clock=ol.CountingClock(), variant='svrg')
    np.random.seed(0)
    po = op.CSMRI(None, H=256, W=256, sample_prob=0.2, snr=20., img=img)
    np.random.seed(1)
    ro = ol.pnp_svrg(po, od.DnCNNDenoiser(wts, 15), 2e3, 2 + 2 * (3 + 5 * 5), 5, 1000, converge_check=False,
                     clock=ol.CountingClock(), variant='svrg')
    ps, pso = np.array(r['psnr_per_iter']), np.array(ro['psnr_per_iter'])
    assert len(ps) == len(pso) == 13
    assert np.abs(ps - pso).max() <= 0.01 + 1e-9, (ps, pso)
    assert np.abs(r['z'] - ro['z']).max() < 5e-4


def test_svrg_graph_replay_equals_eager_loop(api):
    """pnp_svrg with a deterministic clock replays whole outer iterations as hipGraphs (graph=None): identical result
    dict to the eager loop (graph=False) -- iterate, PSNR log, time bookkeeping, RNG stream position, denoiser call
    counter -- for both directions, f32 and f64, incl. a short last outer iteration; and it is what removes the launch
    latency at B = 1."""
    import time
    A, P, D = api
    for dtype in (torch.float32, torch.float64):
        for variant in ('reference', 'svrg'):
            res = []
            for graph in (None, False):
                p = _csmri(P, IMG256, 256, dtype)
                np.random.seed(1)
                d = D.TVDenoiser()
                # 2 + 3 outer x 3 + 5 x 27 ticks: two full outer iterations of 10 and a last one of 7
                r = A.pnp_svrg(p, d, 2e3, 2 + 3 * 3 + 5 * 27 - 1, 10, 1000, verbose=False, converge_check=False,
                               clock=A.CountingClock(), variant=variant, graph=graph)
                res.append((r, d.t, np.random.random()))
            (rg, tg, ug), (re, te, ue) = res
            assert len(rg['psnr_per_iter']) == 1 + 3 + 27 and rg['psnr_per_iter'] == re['psnr_per_iter']
            assert np.array_equal(rg['z'], re['z'])
            assert rg['time_per_iter'] == re['time_per_iter'] and rg['gradient_time'] == re['gradient_time']
            assert rg['denoise_time'] == re['denoise_time'] and tg == te == 27 and ug == ue
    # launch latency: 256 x 256, B = 1, TV prox, true SVRG
    times = {}
    for graph in (None, False):
        p = _csmri(P, IMG256, 256, torch.float32)
        np.random.seed(1)
        n = 200
        torch.cuda.synchronize(); t0 = time.perf_counter()
        A.pnp_svrg(p, D.TVDenoiser(), 2e3, 2 + 3 * (n // 10) + 5 * n - 1, 10, 1000, verbose=False, converge_check=False,
                   clock=A.CountingClock(), variant='svrg', graph=graph)
        torch.cuda.synchronize(); times[graph] = (time.perf_counter() - t0) / n * 1e6
    print(f'drop-in pnp_svrg + TV, 256x256, B = 1: hipGraph {times[None]:.1f} us/inner iteration (incl. host minibatch draws), eager {times[False]:.1f}')
    # the device side alone: replays of one captured outer iteration (T2 = 10 inner iterations)
    from pnp_svrg_amd.algorithms import _SvrgGraph
    p = _csmri(P, IMG256, 256, torch.float32)
    run = _SvrgGraph(p, D.TVDenoiser(), 2e3, 10, 1000, 'svrg', 4096)
    run.upload([[p._select_mb_locs(1000) for _ in range(10)]])
    run.run_outer(0, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        run.graph.replay()
    torch.cuda.synchronize()
    dev_us = (time.perf_counter() - t0) / 500 * 1e6
    print(f'device side of the replayed loop: {dev_us:.1f} us per inner iteration (incl. 1/10 of the full-gradient refresh)')
    assert dev_us < 400      # (at B = 1 the one-workgroup-per-image kernels are bound by a single CU, not by launches)


@pytest.mark.gpu
def test_grad_stoch_minibatch_shortcut_is_only_taken_for_select_mbs_own_array(api):
    """grad_stoch(z, mb) takes select_mb's index list instead of a second flatnonzero when `mb` IS the array select_mb just
    returned (CSMRI.py:66-89 as the loops call them): the same gradient as for an equal copy of it (the general path), and
    an array that was changed afterwards -- one entry removed, one added, or an older draw -- goes the general way too."""
    _, problems, _ = api
    np.random.seed(0)
    p = problems.CSMRI(IMG64, H=64, W=64, sample_prob=0.5, snr=20., dtype=torch.float64)
    z = np.asarray(p.Xinit) + 0.01 * np.random.RandomState(1).rand(p.N)
    old = p.select_mb(200)
    mb = p.select_mb(200)
    g_short = p.grad_stoch(z, mb)
    np.testing.assert_array_equal(g_short, p.grad_stoch(z, mb.copy()))
    np.testing.assert_array_equal(p.grad_stoch(z, old), p.grad_stoch(z, old.copy()))
    assert not np.array_equal(g_short, p.grad_stoch(z, old))
    ys, xs = np.nonzero(mb)
    mb[ys[0], xs[0]] = 0                                        # in place: same object, one entry fewer
    np.testing.assert_array_equal(p.grad_stoch(z, mb), p.grad_stoch(z, mb.copy()))
    assert not np.array_equal(g_short, p.grad_stoch(z, mb))
    free = np.argwhere((p.mask != 0) & (mb == 0))[0]
    mb[free[0], free[1]] = 1                                    # back to 200 entries, a different set
    np.testing.assert_array_equal(p.grad_stoch(z, mb), p.grad_stoch(z, mb.copy()))
