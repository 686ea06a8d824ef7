"""Batched engine == B independent drop-in loops (same minibatches), and bench plumbing."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('variant', ['svrg', 'reference'])
def test_engine_matches_dropin_loop(variant):
    import algorithms
    import problems
    import denoisers
    from oracle import loops as ol
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, TVProx
    B, n, mb, T2, eta, steps = 3, 64, 150, 4, 5e2, 9
    batch = CsmriBatch.synthetic(B, n, n, 0.2, 20.0, seed=5, dtype=torch.float64)
    idx = batch.draw_minibatches(steps, mb, seed=2)
    eng = SvrgEngine(batch, TVProx(), eta, T2, mb, variant=variant)
    for s in range(steps):
        eng.step(idx[s])
    trace = eng.psnr_trace()
    idx_h = idx.cpu().numpy()
    for b in range(B):
        # the same problem through the drop-in API: feed the engine's minibatches via select_mb
        p = problems.CSMRI.__new__(problems.CSMRI)
        problems.Problem.__init__(p, None, n, n, img=batch.xrec[b].cpu().numpy(), dtype=torch.float64)
        p.pname, p.mask, p.M0, p.M = 'csmri', batch.mask_np[b].astype(int), int(batch.M0[b]), n * n
        p.Xrec = batch.xrec[b].cpu().numpy()
        p._xrec_d = batch.xrec[b:b + 1].clone()
        p.Xinit = batch.xinit[b].cpu().numpy().ravel()
        p.Y = None
        p.plan = batch.plan.__class__(n, n, 1, torch.float64)
        p._maskT = batch.maskT[b:b + 1].clone()
        p._yh_full = batch.yh_full[b:b + 1].clone()
        p._selT = torch.empty_like(p._maskT)
        draws = iter(idx_h[:, b])

        def select_mb(size, _d=draws):
            m = np.zeros(n * n, int)
            m[next(_d)] = 1
            return m.reshape(n, n)
        p.select_mb = select_mb
        n_outer = -(-steps // T2)
        r = algorithms.pnp_svrg(p, denoisers.TVDenoiser(), eta, 2 + 3 * n_outer + 5 * steps - 1, T2, mb, verbose=False,
                                converge_check=False, clock=ol.CountingClock(), variant=variant)
        ps = np.array(r['psnr_per_iter'])
        # drop the initial entry and the per-outer entries -> inner-iteration PSNRs
        inner = [v for i, v in enumerate(ps[1:]) if i % (T2 + 1) != 0]
        assert len(inner) == steps
        assert np.abs(np.array(inner) - trace[:, b]).max() <= 1e-9
        np.testing.assert_allclose(r['z'], eng.z[b].cpu().numpy().ravel(), rtol=0, atol=1e-10)


def test_bench_line_smoke():
    import json
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '2', '--warmup', '1', '--batch', '2',
                          '--no-cpu-baseline'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line
    assert line['roofline']['bound'] == 'mfma' and 0 < line['roofline']['frac'] < 1


def test_device_minibatch_draw():
    """pnp_csmri_draw_minibatch: exactly mb ones per problem, inside the mask, deterministic in (seed, step),
    different across steps/problems, and uniform (every sampled location selected ~ mb/M0 of the time)."""
    from pnp_svrg_amd.engine import CsmriBatch
    B, n, mb = 4, 64, 100
    batch = CsmriBatch.synthetic(B, n, n, 0.2, 20.0, seed=9)
    M0 = int(batch.M0[0])
    sel = batch.plan.draw_minibatch(batch.mask_idx, mb, seed=7, step=3)
    s = sel.cpu().numpy()                                     # [B][W][H] transposed
    assert s.dtype == np.uint8 and set(np.unique(s)) <= {0, 1}
    assert (s.reshape(B, -1).sum(1) == mb).all()
    maskT = batch.maskT.cpu().numpy()
    assert (s <= maskT).all()
    assert torch.equal(sel, batch.plan.draw_minibatch(batch.mask_idx, mb, seed=7, step=3))
    assert not torch.equal(sel, batch.plan.draw_minibatch(batch.mask_idx, mb, seed=7, step=4))
    assert not np.array_equal(s[0], s[1])
    # uniformity: 400 draws, per-location frequency ~ Binomial(400, mb/M0)
    cnt = np.zeros((B, n, n), np.int64)
    T = 400
    for t in range(T):
        cnt += batch.plan.draw_minibatch(batch.mask_idx, mb, seed=11, step=t).cpu().numpy()
    p = mb / M0
    f = cnt[maskT == 1] / T
    assert abs(f.mean() - p) < 1e-12 + 1e-9                   # exactly mb per draw
    z = (f - p) / np.sqrt(p * (1 - p) / T)
    assert np.abs(z).max() < 5.5 and abs(z.std() - 1.0) < 0.1
    # edge: mb == M0 selects the whole mask; mb == 1 selects one location
    full = batch.plan.draw_minibatch(batch.mask_idx, M0, seed=1, step=0)
    assert torch.equal(full, batch.maskT)
    one = batch.plan.draw_minibatch(batch.mask_idx, 1, seed=1, step=0)
    assert (one.reshape(B, -1).sum(1) == 1).all()


def _mb_hash_np(seed, step, prob, j):
    """NumPy restatement of the draw's counter-based key (csmri.hip mb_hash: splitmix64 finaliser, high 32 bits)."""
    with np.errstate(over='ignore'):
        x = np.uint64(seed) ^ (np.uint64(step) << np.uint64(40)) ^ (np.uint64(prob) << np.uint64(20)) ^ j.astype(np.uint64)
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(32)).astype(np.uint32)


@pytest.mark.parametrize('n,frac', [(64, 0.3), (256, 0.2), (256, 0.7)])
def test_device_minibatch_draw_known_answer(n, frac):
    """The draw is exactly "the mb smallest hash keys, ties by position" -- recomputed on the host.  256 x 256 at
    70 % sampling (M0 = 45 875 keys) exceeds the LDS key cache and takes the re-hashing build of the kernel; the
    other cases take the cached build."""
    from pnp_svrg_amd import ops
    B, mb, seed, step = 3, 777, 0xDEADBEEFCAFE, 12345
    rng = np.random.default_rng(n)
    M0 = int(round(frac * n * n))
    idx = np.stack([np.sort(rng.choice(n * n, M0, replace=False)) for _ in range(B)]).astype(np.int32)
    plan = ops.CsmriPlan(n, n, B, torch.float32)
    sel = plan.draw_minibatch(torch.from_numpy(idx).cuda(), mb, seed=seed, step=step).cpu().numpy()
    for b in range(B):
        keys = _mb_hash_np(seed, step, b, np.arange(M0))
        order = np.lexsort((np.arange(M0), keys))[:mb]
        want = np.zeros((n, n), np.uint8)
        i = idx[b][order]
        want[i % n, i // n] = 1                                   # transposed selector [W][H]
        assert np.array_equal(sel[b], want)


def test_sweep_runner_single_process():
    """Config-5 shape in one process: images x sampling ratios batched through the engine; every item is
    reconstructed (PSNR improves) and results come back in canonical order with the CSV schema."""
    from pnp_svrg_amd import sweep
    from pnp_svrg_amd.engine import TVProx
    rng = np.random.default_rng(0)
    imgs = []
    for _ in range(2):
        x = rng.random((64, 64))
        p = np.pad(x, 2, mode='wrap')
        imgs.append(sum(p[i:i + 64, j:j + 64] for i in range(5) for j in range(5)) / 25.0)
    items = sweep.make_items(2, [0.2, 0.4, 0.6], [20.0])
    runner = sweep.csmri_svrg_runner(imgs, lambda: TVProx(), eta=5e2, T2=4, mini_batch_size=100, n_inner=12, H=64, W=64)
    res = sweep.run_sweep(items, runner)
    assert [r['id'] for r in res] == list(range(6))
    for r in res:
        assert r['z'].shape == (64, 64) and np.isfinite(r['z']).all()
        assert np.isfinite(r['psnr_final']) and abs(r['loss'] - (r['psnr_init'] - r['psnr_final'])) < 1e-9
    assert res[0]['psnr_final'] > res[0]['psnr_init']          # 20 % sampling: the reconstruction helps
    # more samples -> better reconstruction of the same image
    assert res[2]['psnr_final'] > res[0]['psnr_final']


def test_grid_search_on_device(tmp_path):
    """the reference's per-item hyper-parameter search as a deterministic grid, trials batched over
    items on the engine: the winner per item is the trial with the best final PSNR, equal to running that trial
    alone, and the result file has the reference's row layout."""
    from pnp_svrg_amd import sweep
    from pnp_svrg_amd.engine import TVProx
    rng = np.random.default_rng(1)
    imgs = []
    for _ in range(2):
        x = rng.random((64, 64))
        p = np.pad(x, 2, mode='wrap')
        imgs.append(sum(p[i:i + 64, j:j + 64] for i in range(5) for j in range(5)) / 25.0)
    items = sweep.make_items(2, [0.2, 0.5], [20.0])

    def make_runner(eta, T2):
        return sweep.csmri_svrg_runner(imgs, lambda: TVProx(), eta=eta, T2=T2, mini_batch_size=100, n_inner=12, H=64, W=64)
    grid = {'eta': [1e1, 5e2, 2e3], 'T2': [4, 6]}
    rows = sweep.grid_search(items, make_runner, grid)
    assert [r['id'] for r in rows] == [0, 1, 2, 3]
    for r in rows:
        alone = {a['id']: a for a in make_runner(**r['params'])(items)}[r['id']]
        assert alone['loss'] == r['loss']                                   # deterministic: same trial, same loss
        for params in sweep.grid_points(grid):
            other = {a['id']: a for a in make_runner(**params)(items)}[r['id']]
            assert not (other['loss'] < r['loss'])
    assert any(r['params']['eta'] != 1e1 for r in rows)                     # the tiny step is not the best everywhere
    sweep.write_tuning_csv(str(tmp_path / 'o.csv'), rows, denoiser='TV')
    assert (tmp_path / 'o.csv').read_text().startswith('Results:')


def test_bench_two_ranks_rehearsal():
    """The N > 1 path of bench.py (barrier, MAX over ranks, final gather) with 2 ranks sharing GPU 0 over gloo
    (RCCL needs one GPU per rank; the driver runs the real N = 2/4/8 on a full node)."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PNP_BENCH_ONE_DEVICE='1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'),
                          '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '2', '--backend', 'gloo',
                          '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['problems_total'] == 4 and line['scaling'] == 'weak'
    assert line['value'] > 0 and line['cpu_baseline'] is None


@pytest.mark.parametrize('prox_kind', ['tv', 'dncnn'])
def test_graph_replay_equals_eager(prox_kind):
    """One outer iteration captured in a hipGraph and replayed == the same iterations launched eagerly
    (bit-identical iterate and PSNR log: same kernels, same device-side minibatch draws)."""
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, TVProx, DnCNNProx
    from pnp_svrg_amd.denoisers import random_dncnn_weights
    import time
    B, n, mb, T2 = 2, 64, 100, 5
    mk = (lambda: TVProx()) if prox_kind == 'tv' else (lambda: DnCNNProx(random_dncnn_weights(17, seed=1), 15))
    batch = CsmriBatch.synthetic(B, n, n, 0.2, 20.0, seed=5)
    e1 = SvrgEngine(batch, mk(), 5e2 if prox_kind == 'tv' else 1.0, T2, mb, seed=3)
    for _ in range(3 * T2):
        e1.step()
    e2 = SvrgEngine(batch, mk(), 5e2 if prox_kind == 'tv' else 1.0, T2, mb, seed=3)
    e2.capture()
    assert e2.s == 0 and torch.equal(e2.z, batch.xinit)           # capture leaves the state untouched
    e2.run_outer(3)
    assert e2.s == e1.s == 15
    assert torch.equal(e1.z, e2.z)
    assert np.array_equal(e1.psnr_trace(), e2.psnr_trace())
    # and it is what removes the launch latency at small batch
    torch.cuda.synchronize(); t0 = time.perf_counter(); e2.run_outer(20); torch.cuda.synchronize(); tg = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20 * T2):
        e1.step()
    torch.cuda.synchronize(); te = time.perf_counter() - t0
    print(f'[{prox_kind}] B={B} {n}x{n}: eager {te / (20 * T2) * 1e6:.1f} us/step, graph {tg / (20 * T2) * 1e6:.1f} us/step')


@pytest.mark.parametrize('n,host_idx', [(64, False), (64, True), (256, False)])
def test_fused_tv_engine_equals_plain_engine(n, host_idx):
    """SvrgEngineFusedTV (transposed storage, gradient step + noise estimate + prox + error in one kernel) walks the
    same trajectory as SvrgEngine: same minibatches (device draws or host index lists), iterates equal to ~1e-5,
    PSNR logs within 0.01 dB, eager and hipGraph forms."""
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, SvrgEngineFusedTV, TVProx, make_engine
    B, mb, T2, steps = 3, 150, 5, 15
    batch = CsmriBatch.synthetic(B, n, n, 0.2, 20.0, seed=21)
    plain = SvrgEngine(batch, TVProx(sigma_modifier=1.2), 5e2, T2, mb, seed=4)
    fused = make_engine(batch, TVProx(sigma_modifier=1.2), 5e2, T2, mb, seed=4, fused=True)
    assert isinstance(fused, SvrgEngineFusedTV)
    assert type(make_engine(batch, TVProx(), 5e2, T2, mb)) is SvrgEngine        # opt-in only (measured slower)
    assert torch.equal(fused.z, batch.xinit)
    idx = batch.draw_minibatches(steps, mb, seed=2) if host_idx else None
    for s in range(steps):
        plain.step(None if idx is None else idx[s])
        fused.step(None if idx is None else idx[s])
    assert torch.equal(plain.selT, fused.selT.transpose(1, 2))          # the same minibatch, transposed problem
    d = (plain.z - fused.z).abs().max().item()
    assert d <= 2e-5, d
    assert np.abs(plain.psnr_trace() - fused.psnr_trace()).max() <= 0.01 + 1e-9
    assert plain.psnr_trace()[-1].mean() > plain.psnr_trace()[0].mean()
    assert fused.prox.t == plain.prox.t == steps
    # hipGraph form of the fused engine == its eager form
    g = make_engine(batch, TVProx(sigma_modifier=1.2), 5e2, T2, mb, seed=4, fused=True)
    g.capture()
    assert torch.equal(g.z, batch.xinit)
    g.run_outer(steps // T2)
    if idx is None:
        assert torch.equal(g.z.contiguous(), fused.z.contiguous())
        assert np.array_equal(g.psnr_trace(), fused.psnr_trace())
    # not eligible -> the plain engine
    b64 = CsmriBatch.synthetic(2, 64, 64, 0.2, 20.0, seed=1, dtype=torch.float64)
    assert type(make_engine(b64, TVProx(), 5e2, T2, mb)) is SvrgEngine
    with pytest.raises(ValueError):
        make_engine(b64, TVProx(), 5e2, T2, mb, fused=True)
