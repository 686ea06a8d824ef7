"""Batched engines == B independent drop-in loops (same minibatches), device draws, and bench plumbing."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dropin_csmri(batch, b):
    """Problem b of a CsmriBatch as a drop-in `problems.CSMRI` (same device data, B = 1 plan)."""
    import problems
    n = batch.H
    p = problems.CSMRI.__new__(problems.CSMRI)
    problems.Problem.__init__(p, None, n, n, img=batch.xrec[b].cpu().numpy(), dtype=batch.dtype)
    p.pname, p.mask, p.M0, p.M = 'csmri', batch.mask_np[b].astype(int), int(batch.M0[b]), n * n
    p.Xrec = batch.xrec[b].cpu().numpy()
    p._xrec_d = batch.xrec[b:b + 1].clone()
    p.Xinit = batch.xinit[b].double().cpu().numpy().ravel()
    p.Y = None
    p.plan = batch.plan.__class__(n, n, 1, batch.dtype)
    p._maskT = batch.maskT[b:b + 1].clone()
    p._YT = batch.YT[b:b + 1].clone()
    p._yh_full = batch.yh_full[b:b + 1].clone()
    p._selT = torch.empty_like(p._maskT)
    return p


def feed_minibatches(p, lists, shape):
    """Make p.select_mb return the given index lists, one per call (as 0/1 indicators of `shape`)."""
    it = iter(lists)

    def select_mb(size, _it=it):
        m = np.zeros(int(np.prod(shape)), int)
        m[next(_it)] = 1
        return m.reshape(shape)
    p.select_mb = select_mb


@pytest.mark.parametrize('variant', ['svrg', 'reference'])
def test_engine_matches_dropin_loop(variant):
    import algorithms
    import denoisers
    from oracle import loops as ol
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, TVProx
    B, n, mb, T2, eta, steps = 3, 64, 150, 4, 5e2, 9
    batch = CsmriBatch.synthetic(B, n, n, 0.2, 20.0, seed=5, dtype=torch.float64)
    assert len(set(batch.M0.tolist())) > 1                    # Bernoulli masks: per-problem M0
    idx = batch.draw_minibatches(steps, mb, seed=2)
    eng = SvrgEngine(batch, TVProx(), eta, T2, mb, variant=variant)
    for s in range(steps):
        eng.step(idx[s])
    trace = eng.psnr_trace()
    idx_h = idx.cpu().numpy()
    for b in range(B):
        # the same problem through the drop-in API: feed the engine's minibatches via select_mb
        p = dropin_csmri(batch, b)
        feed_minibatches(p, idx_h[:, b], (n, n))
        n_outer = -(-steps // T2)
        r = algorithms.pnp_svrg(p, denoisers.TVDenoiser(), eta, 2 + 3 * n_outer + 5 * steps - 1, T2, mb, verbose=False,
                                converge_check=False, clock=ol.CountingClock(), variant=variant)
        ps = np.array(r['psnr_per_iter'])
        # drop the initial entry and the per-outer entries -> inner-iteration PSNRs
        inner = [v for i, v in enumerate(ps[1:]) if i % (T2 + 1) != 0]
        assert len(inner) == steps
        assert np.abs(np.array(inner) - trace[:, b]).max() <= 1e-9
        np.testing.assert_allclose(r['z'], eng.z[b].cpu().numpy().ravel(), rtol=0, atol=1e-10)


@pytest.mark.parametrize('algo', ['gd', 'sgd', 'sarah', 'saga'])
def test_other_engines_match_dropin_loops(algo):
    """GdEngine / SgdEngine / SarahEngine / SagaEngine over a CsmriBatch (Bernoulli masks, float64) walk the
    trajectories of the golden-pinned drop-in loops pnp_gd / pnp_sgd / pnp_sarah / pnp_saga fed the same minibatches
    (and the same replaced-row stream for SAGA): identical rounded PSNR logs, iterates to 1e-10."""
    import algorithms
    import denoisers
    from oracle import loops as ol
    from pnp_svrg_amd.engine import CsmriBatch, TVProx, make_engine
    B, n, mb, T2, eta, steps, hist, decay = 3, 64, 150, 4, 5e2, 10, 5, 0.95
    batch = CsmriBatch.synthetic(B, n, n, 0.25, 20.0, seed=8, dtype=torch.float64)
    idx = batch.draw_minibatches(steps + 1, mb, seed=3)
    idx_h = idx.cpu().numpy()
    np.random.seed(77)
    rs = [np.random.choice(hist, 1).item() for _ in range(steps)]
    if algo == 'saga':
        eng = make_engine(batch, TVProx(), eta, T2, mb, lr_decay=decay, algorithm='saga', hist_size=hist, idx0=idx[0])
        for s in range(steps):
            eng.step(idx[s + 1], r=rs[s])
    else:
        eng = make_engine(batch, TVProx(), eta, T2, mb, lr_decay=decay, algorithm=algo)
        for s in range(steps):
            eng.step() if algo == 'gd' else eng.step(idx[s])
    trace = eng.psnr_trace()
    for b in range(B):
        p = dropin_csmri(batch, b)
        d = denoisers.TVDenoiser()
        kw = dict(verbose=False, converge_check=False, clock=ol.CountingClock(), lr_decay=decay)
        if algo == 'gd':
            r = algorithms.pnp_gd(p, d, eta, 6 * steps - 3, **kw)
            got = np.array(r['psnr_per_iter'])[1:]
        elif algo == 'sgd':
            feed_minibatches(p, idx_h[:, b], (n, n))
            r = algorithms.pnp_sgd(p, d, eta, 5 * steps - 2, mb, **kw)
            got = np.array(r['psnr_per_iter'])[1:]
        elif algo == 'saga':
            feed_minibatches(p, idx_h[:, b], (n, n))
            np.random.seed(77)
            r = algorithms.pnp_saga(p, d, eta, 5 * steps - 1, mb, hist_size=hist, **kw)
            got = np.array(r['psnr_per_iter'])[1:]
        else:
            feed_minibatches(p, idx_h[:, b], (n, n))
            o, j = (steps - 1) // T2, (steps - 1) % T2
            r = algorithms.pnp_sarah(p, d, eta, 1 + o * (5 + 5 * T2) + 5 + 5 * j + 1, T2, mb, **kw)
            got = np.array(r['psnr_per_iter'])                  # outer prox entries included, like the engine's log
        assert len(got) == trace.shape[0], (len(got), trace.shape)
        assert np.abs(got - trace[:, b]).max() <= 1e-9, (algo, got, trace[:, b])
        np.testing.assert_allclose(r['z'], eng.z[b].cpu().numpy().ravel(), rtol=0, atol=1e-10)


def test_log_ring_is_chronological():
    """More prox evaluations than log rows: psnr_trace() returns the LAST n_log of them in order (trace[-1] is the
    latest step)."""
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, TVProx
    batch = CsmriBatch.synthetic(2, 64, 64, 0.2, 20.0, seed=5)
    big = SvrgEngine(batch, TVProx(), 5e2, 3, 100, seed=3)
    small = SvrgEngine(batch, TVProx(), 5e2, 3, 100, seed=3, n_log=4)
    for _ in range(7):
        big.step()
        small.step()
    assert np.array_equal(small.psnr_trace(), big.psnr_trace()[-4:])


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
    rf = line['roofline']
    assert rf['bound'] == 'mfma' and 0 < rf['frac'] < 1
    # frac is EXECUTED matrix-core work over the peak; the algorithmic rate is reported beside it
    assert rf['winograd_reduction'] == 4.0                       # F(4x4,3x3) is the default conv kernel at H % 8 == 0, W % 64 == 0
    assert abs(rf['algorithmic_tflops'] / rf['achieved'] - 4.0) < 0.01
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3


def test_bench_default_line_has_secondary_configs():
    """The default configuration (B = 120) carries configs 2 and 4 as `secondary` with their own rooflines, and the
    headline `frac` stays a fraction at the default batch."""
    import json
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '3', '--warmup', '1', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert 0 < line['roofline']['frac'] < 1 and line['config']['batch_per_gpu'] == 120
    sec = line['secondary']
    assert sec['tv']['roofline']['bound'] == 'hbm' and 0 < sec['tv']['roofline']['frac'] < 1 and sec['tv']['config']['batch_per_gpu'] == 1024
    assert sec['saga-nlm']['roofline']['bound'] == 'valu' and sec['saga-nlm']['value'] > 0
    assert sec['saga-nlm']['roofline']['saga_table_update']['bound'] == 'hbm'
    assert sec['tv']['psnr_db']['after_timed_steps_mean'] > sec['tv']['psnr_db']['initial_mean']


def _mix64(x):
    with np.errstate(over='ignore'):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _mb_keys_np(seed, step, prob, pos):
    """NumPy restatement of the draw's counter-based key (include/pnp_hip.h, csrc/draw.h): the stream state absorbs
    seed, step and problem one at a time (splitmix64); a position's key is a 32-bit mixer of position ^ lo32(state),
    xor hi32(state)."""
    with np.errstate(over='ignore'):
        st = _mix64(_mix64(_mix64(np.uint64(seed)) + np.uint64(step)) + np.uint64(prob))
        x = (np.uint32(int(st) & 0xFFFFFFFF) ^ pos.astype(np.uint32)).astype(np.uint32)
        x ^= x >> np.uint32(16)
        x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
        x ^= x >> np.uint32(15)
        x = (x * np.uint32(0x846ca68b)).astype(np.uint32)
        x ^= x >> np.uint32(16)
        return (x ^ np.uint32(int(st) >> 32)).astype(np.uint32)


def test_device_minibatch_draw():
    """pnp_csmri_draw_minibatch: exactly mb ones per problem, inside the mask, deterministic in (seed, step),
    different across steps/problems, and uniform (every sampled location selected ~ mb/M0 of the time)."""
    from pnp_svrg_amd.engine import CsmriBatch
    B, n, mb = 4, 64, 100
    batch = CsmriBatch.synthetic(B, n, n, 0.2, 20.0, seed=9)
    sel = batch.plan.draw_minibatch(batch.bits, mb, seed=7, step=3)
    s = sel.cpu().numpy()                                     # [B][W][H] transposed
    assert s.dtype == np.uint8 and set(np.unique(s)) <= {0, 1}
    assert (s.reshape(B, -1).sum(1) == mb).all()
    maskT = batch.maskT.cpu().numpy()
    assert (s <= maskT).all()
    assert torch.equal(sel, batch.plan.draw_minibatch(batch.bits, mb, seed=7, step=3))
    assert not torch.equal(sel, batch.plan.draw_minibatch(batch.bits, mb, seed=7, step=4))
    assert not np.array_equal(s[0], s[1])
    # uniformity: 400 draws, per-location frequency ~ Binomial(400, mb/M0_b)
    cnt = np.zeros((B, n, n), np.int64)
    T = 400
    for t in range(T):
        cnt += batch.plan.draw_minibatch(batch.bits, mb, seed=11, step=t).cpu().numpy()
    for b in range(B):
        p = mb / batch.M0[b]
        f = cnt[b][maskT[b] == 1] / T
        assert abs(f.mean() - p) < 1e-9                       # exactly mb per draw
        z = (f - p) / np.sqrt(p * (1 - p) / T)
        assert np.abs(z).max() < 5.5 and abs(z.std() - 1.0) < 0.15
    # edge: mb >= M0 selects the whole mask; mb == 1 selects one location
    full = batch.plan.draw_minibatch(batch.bits, int(batch.M0.max()), seed=1, step=0)
    assert torch.equal(full[int(np.argmax(batch.M0))], batch.maskT[int(np.argmax(batch.M0))])
    assert torch.equal(batch.plan.draw_minibatch(batch.bits, n * n, seed=1, step=0), batch.maskT)
    one = batch.plan.draw_minibatch(batch.bits, 1, seed=1, step=0)
    assert (one.reshape(B, -1).sum(1) == 1).all()


@pytest.mark.parametrize('n,frac', [(64, 0.3), (128, 0.5), (256, 0.2), (256, 0.7)])
def test_device_minibatch_draw_known_answer(n, frac):
    """The draw is exactly "the mb smallest (key, position) pairs" -- recomputed on the host from the published key
    construction (include/pnp_hip.h).  Problems of one call have different M0."""
    from pnp_svrg_amd import ops
    B, mb, seed, step = 3, 777, 0xDEADBEEFCAFE, 12345
    rng = np.random.default_rng(n)
    plan = ops.CsmriPlan(n, n, B, torch.float32)
    mask = (rng.random((B, n, n)) < np.array([frac, 0.8 * frac, 1.2 * frac])[:, None, None]).astype(np.uint8)
    bits = plan.pack_mask(plan.sel_from_dense(torch.from_numpy(mask).cuda()))
    mbd = plan.draw_thresholds(bits, mb, seed, step, nsteps=2)
    for t in range(2):
        sel = plan.sel_from_thresholds(bits, mbd[t]).cpu().numpy()
        for b in range(B):
            pos = np.flatnonzero(mask[b])                         # flat row-major positions, ascending
            keys = _mb_keys_np(seed, step + t, b, pos)
            order = np.lexsort((pos, keys))[:mb]
            want = np.zeros(n * n, np.uint8)
            want[pos[order]] = 1
            assert np.array_equal(sel[b], want.reshape(n, n).T)   # transposed selector [W][H]
            T, P = keys[order[-1]], pos[order[-1]]
            d = mbd[t, b].cpu().numpy().view(np.uint32)           # {state lo, state hi, T, P}
            assert (int(d[2]), int(d[3])) == (int(T), int(P))


def test_draw_fast_path_equals_general_select(monkeypatch):
    """The draw kernel's one-sweep fast path (window around the expected threshold) and its general radix select
    (PNP_DRAW_NO_FAST=1) produce the same descriptors and bits, masked and unmasked, incl. mb close to M0."""
    from pnp_svrg_amd import ops
    n, B = 256, 3
    rng = np.random.default_rng(4)
    plan = ops.CsmriPlan(n, n, B, torch.float32)
    mask = (rng.random((B, n, n)) < np.array([0.2, 0.05, 0.6])[:, None, None]).astype(np.uint8)
    bits = plan.pack_mask(plan.sel_from_dense(torch.from_numpy(mask).cuda()))
    m0min = int(mask.reshape(B, -1).sum(1).min())
    for mb in (1, 1000, m0min - 1, m0min):
        out = {}
        for fast in (True, False):
            if fast:
                monkeypatch.delenv('PNP_DRAW_NO_FAST', raising=False)
            else:
                monkeypatch.setenv('PNP_DRAW_NO_FAST', '1')
            sb = torch.zeros((2, B, n, n // 32), dtype=torch.int32, device='cuda')
            mbd = plan.draw_thresholds(bits, mb, seed=99, step0=3, nsteps=2, selbits=sb)
            gen = ops.draw_thresholds(5000, B, min(mb, 4999), seed=99, step0=3, nsteps=2)
            out[fast] = (mbd.clone(), sb.clone(), gen.clone())
        for a, b in zip(out[True], out[False]):
            assert torch.equal(a, b), mb
    monkeypatch.delenv('PNP_DRAW_NO_FAST', raising=False)


def test_device_draw_seeds_are_independent():
    """streams of different seeds / problems must be unrelated.  Two problems with the SAME mask: the
    seed-0 and seed-1 minibatches overlap like independent draws (~ mb^2 / M0, hypergeometric), and the field
    boundaries do not alias: (seed = 2^20, problem 0) != (seed = 0, problem 1), (step = 2^24 + 1) != (step = 1)."""
    from pnp_svrg_amd import ops
    n, mb = 256, 1000
    rng = np.random.default_rng(0)
    m = (rng.random((n, n)) < 0.2).astype(np.uint8)
    mask = np.stack([m, m])
    plan = ops.CsmriPlan(n, n, 2, torch.float32)
    bits = plan.pack_mask(plan.sel_from_dense(torch.from_numpy(mask).cuda()))
    M0 = int(m.sum())
    draw = lambda seed, step: plan.draw_minibatch(bits, mb, seed=seed, step=step).cpu().numpy().astype(np.int64)
    a, b = draw(0, 5), draw(1, 5)
    mean, sd = mb * mb / M0, np.sqrt(mb * (mb / M0) * (1 - mb / M0) * (M0 - mb) / (M0 - 1))
    for p in range(2):
        ov = int((a[p] * b[p]).sum())
        assert abs(ov - mean) < 6 * sd, (ov, mean, sd)            # the old XOR construction gave a permuted copy
    ov01 = int((a[0] * a[1]).sum())                               # same seed, same mask, different problem
    assert abs(ov01 - mean) < 6 * sd
    c = draw(2 ** 20, 5)
    assert not np.array_equal(c[0], a[1])
    assert not np.array_equal(draw(0, 2 ** 24 + 1)[0], draw(0, 1)[0])


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
    # the runner replays whole outer iterations as hipGraphs (device-drawn minibatches): same bits as eager stepping
    eager = sweep.run_sweep(items, sweep.csmri_svrg_runner(imgs, lambda: TVProx(), eta=5e2, T2=4, mini_batch_size=100, n_inner=12,
                                                           H=64, W=64, graph=False))
    for a, b in zip(res, eager):
        assert np.array_equal(a['z'], b['z']) and a['psnr_final'] == b['psnr_final']


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


def test_one_kernel_iteration_equals_four_kernels():
    """pnp_csmri_svrg_step (csrc/csmri_fused.hip: SVRG step + noise estimate + TV prox + error, image register-resident)
    against pnp_csmri_grad_sel followed by pnp_prox_tv: same stepped image / noise estimate / prox to fp32 rounding
    (the FFTs are the same building blocks, compiled with and without FMA contraction), in place and out of place,
    with and without the prox (denoise = 0 is what the DnCNN prox follows)."""
    from pnp_svrg_amd import ops
    from pnp_svrg_amd.engine import CsmriBatch
    B, mb = 5, 1000
    batch = CsmriBatch.synthetic(B, 256, 256, 0.2, 20.0, seed=31)
    p = batch.plan
    rng = np.random.default_rng(0)
    z = batch.xinit.clone()
    w = (batch.xinit + torch.from_numpy(0.05 * rng.standard_normal((B, 256, 256))).float().cuda()).contiguous()
    mu = torch.from_numpy(1e-4 * rng.standard_normal((B, 256, 256))).float().cuda()
    selbits = torch.empty((1, B, 256, 8), dtype=torch.int32, device='cuda')
    p.draw_thresholds(batch.bits, mb, seed=3, step0=7, nsteps=1, selbits=selbits)
    lr = 2e3
    kw = dict(alpha=-lr / mb, beta=1.0, c1=z, gamma=-lr, c2=mu)
    stepped = p.grad(z, bits=selbits[0], b=w, **kw)
    want, want_sse, want_sig = ops.prox_tv(stepped, xrec=batch.xrec, sigma_modifier=1.3)
    got, sse, sig = p.svrg_step(z, w, selbits[0], xrec=batch.xrec, sigma_modifier=1.3,
                                sse=torch.empty(B, dtype=torch.float64, device='cuda'), **kw)
    assert not torch.equal(want, stepped)
    assert (sig - want_sig).abs().max().item() <= 1e-6 * want_sig.abs().max().item()
    assert (got - want).abs().max().item() <= 2e-5
    np.testing.assert_allclose(sse.cpu().numpy(), want_sse.cpu().numpy(), rtol=1e-4)
    # stop after the noise estimate: the stepped image itself
    raw, _, sig0 = p.svrg_step(z, w, selbits[0], denoise=False, **kw)
    assert (raw - stepped).abs().max().item() <= 2e-6
    assert (sig0 - want_sig).abs().max().item() <= 1e-6 * want_sig.abs().max().item()
    # in place (out aliases a and c1, as the engine calls it); per-problem scale; no second operand
    z2 = z.clone()
    p.svrg_step(z2, w, selbits[0], alpha=-lr / mb, beta=1.0, c1=z2, gamma=-lr, c2=mu, out=z2, xrec=batch.xrec, sigma_modifier=1.3)
    assert torch.equal(z2, got)
    av = torch.tensor([0.5, 2.0, 1.0, 0.25, 3.0], device='cuda')
    g1, _, _ = p.svrg_step(z, None, batch.bits, alpha=1e-4, alpha_vec=av, denoise=False)
    g2 = p.grad(z, bits=batch.bits, alpha=1e-4, alpha_vec=av)
    assert (g1 - g2).abs().max().item() <= 2e-6 * max(1.0, g2.abs().max().item())
    # gradient-only mode with the packed data term (what grad_full of a large batch takes): force it on this small batch
    import os
    from pnp_svrg_amd import ops as _ops
    os.environ['PNP_CSMRI_FUSED_MIN_BATCH'] = '1'
    try:
        pf = _ops.CsmriPlan(256, 256, B, torch.float32)
    finally:
        del os.environ['PNP_CSMRI_FUSED_MIN_BATCH']
    gf1 = pf.grad(z, bits=batch.bits, yh=batch.yh_full, alpha=0.7, alpha_vec=batch.inv_m0, beta=0.5, c1=w)
    gf2 = p.grad(z, bits=batch.bits, yh=batch.yh_full, alpha=0.7, alpha_vec=batch.inv_m0, beta=0.5, c1=w)
    # (different kernels; since the complex products spell their fused multiply-adds out, common.h, they may even agree bit for bit)
    assert (gf1 - gf2).abs().max().item() <= 2e-6 * max(1.0, gf2.abs().max().item())
    # every combination of epilogue operands (the kernel is instantiated per operand count; a lone c2 takes the first slot),
    # gradient-only form against the streaming kernels, whole iteration against gradient + prox
    for kw2 in (dict(), dict(beta=0.5, c1=w), dict(gamma=-0.25, c2=mu), dict(beta=1.0, c1=z, gamma=-lr, c2=mu)):
        for bb in (None, w):
            ga = pf.grad(z, bits=selbits[0], b=bb, alpha=-lr / mb, **kw2)
            gb = p.grad(z, bits=selbits[0], b=bb, alpha=-lr / mb, **kw2)
            assert (ga - gb).abs().max().item() <= 2e-6 * max(1.0, gb.abs().max().item()), (kw2.keys(), bb is None)
        want2, want2_sse, _ = ops.prox_tv(p.grad(z, bits=selbits[0], b=w, alpha=-lr / mb, **kw2), xrec=batch.xrec)
        got2, sse2, _ = p.svrg_step(z, w, selbits[0], alpha=-lr / mb, xrec=batch.xrec,
                                    sse=torch.empty(B, dtype=torch.float64, device='cuda'), **kw2)
        assert (got2 - want2).abs().max().item() <= 2e-5, kw2.keys()
        np.testing.assert_allclose(sse2.cpu().numpy(), want2_sse.cpu().numpy(), rtol=1e-4)


@pytest.mark.parametrize('prox_kind', ['tv', 'dncnn'])
def test_fused_engine_equals_unfused(prox_kind):
    """SvrgEngine with the one-kernel iteration (default for f32 256 x 256 CSMRI) walks the trajectory of the
    four-kernel engine: device draws or host index lists, eager and hipGraph forms."""
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, TVProx, DnCNNProx
    from pnp_svrg_amd.denoisers import random_dncnn_weights
    B, mb, T2, steps = 3, 1000, 4, 8
    mk = (lambda: TVProx(sigma_modifier=1.1)) if prox_kind == 'tv' else (lambda: DnCNNProx(random_dncnn_weights(17, seed=1), 15))
    eta = 2e3 if prox_kind == 'tv' else 1.0
    batch = CsmriBatch.synthetic(B, 256, 256, 0.2, 20.0, seed=13)
    for host in (False, True):
        ef = SvrgEngine(batch, mk(), eta, T2, mb, seed=4, fused=True)     # (automatic from B = 192 on)
        eu = SvrgEngine(batch, mk(), eta, T2, mb, seed=4, fused=False)
        assert ef.fused and not eu.fused
        idx = batch.draw_minibatches(steps, mb, seed=2) if host else None
        for s in range(steps):
            ef.step(None if idx is None else idx[s])
            eu.step(None if idx is None else idx[s])
        assert (ef.z - eu.z).abs().max().item() <= 5e-5 * max(1.0, eu.z.abs().max().item())
        assert np.abs(ef.psnr_trace() - eu.psnr_trace()).max() <= 0.01 + 1e-9
    g = SvrgEngine(batch, mk(), eta, T2, mb, seed=4, fused=True)
    g.capture()
    g.run_outer(steps // T2)
    e = SvrgEngine(batch, mk(), eta, T2, mb, seed=4, fused=True)
    for s in range(steps):
        e.step()
    assert torch.equal(g.z, e.z) and np.array_equal(g.psnr_trace(), e.psnr_trace())
    assert type(g.prox).__name__ != 'TVProx' or g.prox.t == e.prox.t == steps
    assert not SvrgEngine(batch, mk(), eta, T2, mb).fused              # small batch: the streaming kernels


@pytest.mark.parametrize('prox_kind', ['tv', 'dncnn'])
def test_folded_outer_refresh_is_bit_identical(prox_kind, monkeypatch):
    """pnp_csmri_svrg_outer_step -- mu = grad_full(z), w = z and the first inner iteration of the outer iteration in ONE
    kernel (algorithms/pnp_svrg.py:32-57 at j = 0, where gs(z) - gs(w) == 0) -- against the three launches it replaces
    (one-kernel gradient, copy, one-kernel iteration): identical mu, w, iterate and PSNR log, eager and as a hipGraph."""
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, TVProx, DnCNNProx
    from pnp_svrg_amd.denoisers import random_dncnn_weights
    monkeypatch.setenv('PNP_CSMRI_FUSED_MIN_BATCH', '1')           # grad_full of this small batch through the one-kernel gradient
    B, mb, T2, steps = 3, 1000, 4, 9
    mk = (lambda: TVProx(sigma_modifier=1.1)) if prox_kind == 'tv' else (lambda: DnCNNProx(random_dncnn_weights(17, seed=1), 15))
    eta = 2e3 if prox_kind == 'tv' else 1.0
    batch = CsmriBatch.synthetic(B, 256, 256, 0.2, 20.0, seed=17)
    ef = SvrgEngine(batch, mk(), eta, T2, mb, seed=4, fused=True, fold_outer=True)
    eu = SvrgEngine(batch, mk(), eta, T2, mb, seed=4, fused=True, fold_outer=False)
    for s in range(steps):
        ef.step()
        eu.step()
        if s % T2 == 0:
            assert torch.equal(ef.mu, eu.mu) and torch.equal(ef.w, eu.w)
            assert ef.mu.abs().max().item() > 0
        assert torch.equal(ef.z, eu.z), s
    assert np.array_equal(ef.psnr_trace(), eu.psnr_trace())
    assert type(ef.prox).__name__ != 'TVProx' or ef.prox.t == eu.prox.t == steps
    g = SvrgEngine(batch, mk(), eta, T2, mb, seed=4, fused=True)
    assert g.fold_outer
    g.capture()
    g.run_outer(2)
    e = SvrgEngine(batch, mk(), eta, T2, mb, seed=4, fused=True, fold_outer=False)
    for s in range(2 * T2):
        e.step()
    assert torch.equal(g.z, e.z) and np.array_equal(g.psnr_trace(), e.psnr_trace())
    # host-fed minibatches take the same route (the minibatch of step 0 is drawn and ignored)
    idx = batch.draw_minibatches(T2 + 1, mb, seed=2)
    h1 = SvrgEngine(batch, mk(), eta, T2, mb, seed=4, fused=True, fold_outer=True)
    h2 = SvrgEngine(batch, mk(), eta, T2, mb, seed=4, fused=True, fold_outer=False)
    for s in range(T2 + 1):
        h1.step(idx[s]); h2.step(idx[s])
    assert torch.equal(h1.z, h2.z) and np.array_equal(h1.psnr_trace(), h2.psnr_trace())


def test_outer_iteration_in_one_launch_is_bit_identical():
    """pnp_csmri_svrg_outer_iteration (`SvrgEngine.run_outer`, one draw + ONE kernel per outer iteration: the workgroup that owns a
    problem runs the folded refresh and its T2 inner iterations back to back) == the same iterations stepped one launch each:
    identical iterate, w, mu, PSNR log and prox counter; across a wrap of the log ring; with a decaying step size."""
    from pnp_svrg_amd.engine import CsmriBatch, SvrgEngine, TVProx
    B, mb, T2 = 3, 1000, 4
    batch = CsmriBatch.synthetic(B, 256, 256, 0.2, 20.0, seed=23)
    for decay, n_log in ((1.0, 4096), (0.9, 6)):
        e1 = SvrgEngine(batch, TVProx(sigma_modifier=1.2), 2e3, T2, mb, lr_decay=decay, seed=5, fused=True, n_log=n_log)
        e2 = SvrgEngine(batch, TVProx(sigma_modifier=1.2), 2e3, T2, mb, lr_decay=decay, seed=5, fused=True, n_log=n_log)
        assert e1.outer_kernel_ok()
        e1.run_outer(3)
        for _ in range(3 * T2):
            e2.step()
        assert e1.s == e2.s == 3 * T2 and e1.prox.t == e2.prox.t
        assert torch.equal(e1.z, e2.z) and torch.equal(e1.w, e2.w) and torch.equal(e1.mu, e2.mu)
        assert np.array_equal(e1.psnr_trace(), e2.psnr_trace())
        assert torch.equal(e1.prox.sig, e2.prox.sig)
        # and on: stepping continues from a one-launch run, a one-launch run from stepping
        e1.step(); e2.step()
        assert torch.equal(e1.z, e2.z)
    with pytest.raises(ValueError):
        e1.run_outer(1, one_launch=True)                        # 13 steps done: not a multiple of T2
    # host-fed selectors cannot take this route
    e3 = SvrgEngine(batch, TVProx(), 2e3, T2, mb, seed=5, fused=True)
    e3.step(batch.draw_minibatches(1, mb, seed=2)[0])
    assert not e3.outer_kernel_ok()
