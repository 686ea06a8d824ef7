"""Sharding of independent reconstructions over the GPUs of one node (BASELINE config 5).

Mirrors the work decomposition of reference script_diff_sampratio_set12.py:113-146 /
script_diff_snr_set12.py (one work item = image x sampling ratio x SNR x seed, all independent;
the reference maps them over a multiprocessing.Pool): items are dealt round-robin to ranks, one
process per GPU, every rank batches ITS items through the engine, and the only collective is the
final gather of the (small) results -- RCCL over xGMI when the group is NCCL, gloo in CPU tests.
There is no data-path collective to overlap: a 256x256 reconstruction is never split across GPUs.
"""
import csv
import os

import numpy as np
import torch
import torch.distributed as dist


def make_items(n_images, alphas, snrs, seeds=(0,)):
    """Canonical work-item list (same nesting order as the reference's loops: image, alpha, snr)."""
    items = []
    for img in range(n_images):
        for a in alphas:
            for s in snrs:
                for sd in seeds:
                    items.append({'id': len(items), 'image': img, 'alpha': float(a), 'snr': float(s), 'seed': int(sd)})
    return items


def shard(items, rank, world):
    """Static round-robin: item i -> rank i % world (items of equal cost; no exchange afterwards).  A rank's items
    are batched TOGETHER whatever their sampling ratio (per-problem 1/M0 and minibatch thresholds in the engine), so
    120 items on 8 ranks are 8 batches of 15, not 40 batches of 3."""
    return [it for it in items if it['id'] % world == rank]


def gather_results(local, dst=0, group=None):
    """Final gather of per-item results (list of dicts with an 'id').  Returns the full list sorted by
    id on rank `dst`, None elsewhere.  Single-process: returns the sorted local list."""
    if not (dist.is_available() and dist.is_initialized()):
        return sorted(local, key=lambda r: r['id'])
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out = [None] * world if rank == dst else None
    dist.gather_object(local, out, dst=dst, group=group)
    if rank != dst:
        return None
    return sorted((r for part in out for r in part), key=lambda r: r['id'])


def gather_device(z_local, meta_local, n_items, dst=0, group=None):
    """The final gather as two tensor collectives (RCCL over xGMI when the group is NCCL): the reconstructions
    z_local [n_local, H, W] (device tensor, any float dtype) and one small float64 row per item meta_local [n_local, K] whose
    first column is the item id.  Ranks hold different numbers of items (round-robin shards), so rows are padded to
    ceil(n_items / world).  Returns (z [n_items, H, W], meta [n_items, K]) in id order on rank `dst` (CPU tensors), None
    elsewhere.  Single-process: the local data, sorted."""
    meta_local = torch.as_tensor(meta_local, dtype=torch.float64).reshape(z_local.shape[0], -1)
    if not (dist.is_available() and dist.is_initialized()):
        meta_local = meta_local.cpu()
        order = torch.argsort(meta_local[:, 0])
        return z_local.detach().cpu()[order], meta_local[order]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    on_dev = dist.get_backend(group) == 'nccl'
    dev = z_local.device if on_dev else torch.device('cpu')
    cap = -(-n_items // world)
    zp = torch.zeros((cap,) + tuple(z_local.shape[1:]), dtype=z_local.dtype, device=dev)
    mp = torch.full((cap, meta_local.shape[1]), -1.0, dtype=torch.float64, device=dev)
    n = z_local.shape[0]
    zp[:n] = z_local.to(dev)
    mp[:n] = meta_local.to(dev)
    zs = [torch.empty_like(zp) for _ in range(world)] if rank == dst else None
    ms = [torch.empty_like(mp) for _ in range(world)] if rank == dst else None
    dist.gather(zp, zs, dst=dst, group=group)
    dist.gather(mp, ms, dst=dst, group=group)
    if rank != dst:
        return None
    z, m = torch.cat(zs).cpu(), torch.cat(ms).cpu()
    keep = m[:, 0] >= 0
    z, m = z[keep], m[keep]
    order = torch.argsort(m[:, 0])
    assert order.numel() == n_items, 'every item exactly once'
    return z[order], m[order]


def run_sweep(items, runner, group=None):
    """Run `runner(my_items) -> [result dict per item]` on this rank's shard and gather on rank 0."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    mine = shard(items, rank, world)
    res = runner(mine) if mine else []
    assert [r['id'] for r in res] == [it['id'] for it in mine], 'runner must return one result per item, in order'
    return gather_results(res, 0, group)


def _norm01(img):
    x = np.asarray(img, np.float64)
    return (x - x.min()) / (x.max() - x.min())


def _csmri_item_generator(img, it, H, W):
    """One work item's data from a Generator stream keyed by the item (fast path of the sweeps): Bernoulli mask like
    the reference (problems/CSMRI.py:43-45), masked spectrum + real noise on the support (:29-33), |ifft2| init."""
    rng = np.random.default_rng(1000003 * it['seed'] + it['id'])
    x = _norm01(img)
    mk = (rng.random((H, W)) < it['alpha']).astype(np.uint8)
    Y0 = mk * np.fft.fft2(x)
    sig = np.sqrt(np.linalg.norm(Y0.ravel()) / 10 ** (it['snr'] / 10) / H / W)
    Y = Y0 + mk * rng.normal(0, sig, (H, W))
    xi = np.absolute(np.fft.ifft2(Y))
    return x, mk, Y, ((xi - xi.min()) / (xi.max() - xi.min())).ravel()


def _minimal_kernel(H, W, kernel):
    """The reference's "Minimal" / "Identity" blur vectors (problems/DeblurSR.py:80-93), already divided by N."""
    Bk = np.zeros((H, W))
    Bk[0, 0] = 1
    if kernel == 'Minimal':
        Bk[H // 2, H // 2] = Bk[H // 2, H // 3] = Bk[H // 2, H // 4] = 1
        Bk /= 4
    elif kernel != 'Identity':
        raise ValueError(f'kernel {kernel!r}: "Minimal" or "Identity" (generator seeding; legacy seeding takes what problems.Deblur takes)')
    return Bk.ravel() / (H * W)


PROBLEMS = ('csmri', 'deblur', 'pr')
ALGORITHMS = ('gd', 'sgd', 'svrg', 'saga', 'sarah')
_REF_NAMES = {'csmri': 'CSMRI', 'deblur': 'DeblurSR', 'pr': 'PR', 'tv': 'TV', 'nlm': 'NLM', 'dncnn': 'CNN'}


def deblur_scale_percent(alpha):
    """item alpha (fraction of the full-resolution measurements, 0.1 .. 1.0) -> Deblur's scale_percent, as the reference's
    get_problem does with its alpha in 1 .. 10 (script_diff_sampratio_set12.py:45-46: int(alpha * 10))."""
    return int(round(alpha * 100))


def pr_num_meas(alpha, H, W):
    """item alpha (measurements per pixel) -> PhaseRetrieval's num_meas (script_diff_sampratio_set12.py:47-48)."""
    return int(alpha * H * W)


def make_prox(denoiser, **kw):
    """'tv' | 'nlm' -> a fresh engine prox (denoisers/TV.py, NLM.py semantics); a callable is called (e.g. a DnCNNProx factory)."""
    from .engine import TVProx, NLMProx
    if callable(denoiser):
        return denoiser()
    if denoiser == 'tv':
        return TVProx(**kw)
    if denoiser == 'nlm':
        return NLMProx(**kw)
    raise ValueError(f'unknown denoiser {denoiser!r}: "tv", "nlm" or a factory')


def make_runner(images, problem='csmri', algorithm='svrg', denoiser='tv', *, eta, n_inner, mini_batch_size=None, T2=None,
                hist_size=50, H=256, W=256, dtype=torch.float32, max_batch=128, seeding='generator', variant='svrg', run_seed=1,
                keep_trace=False, graph=True, kernel='Minimal', lr_decay=1.0, denoiser_kwargs=None):
    """Runner for `run_sweep` / `grid_search` over any cell of the reference's sweep (script_diff_sampratio_set12.py:23-25,
    41-51, 64-131): problem in {'csmri', 'deblur', 'pr'} x algorithm in {'gd', 'sgd', 'svrg', 'saga', 'sarah'} x denoiser in
    {'tv', 'nlm', factory}; `n_inner` inner iterations (prox evaluations of the stepped iterate) per item, hyper-parameters
    eta, mini_batch_size, T2 (svrg, sarah), hist_size (saga) -- the keys a search grid varies.

    A rank's items run as batches on the engines of `engine.py`: CSMRI items of ANY mix of sampling ratios together (per-problem
    1/M0 and minibatch thresholds); Deblur / PR items are grouped by alpha, which fixes the operator's shape (scale_percent =
    100 alpha; num_meas = alpha H W).
    seeding='generator': per-item data from a Generator stream keyed by the item, minibatches drawn on the device;
    seeding='legacy'   : per item exactly the reference's RNG use -- np.random.seed(item seed), the problem constructor's draws
                         in its order, np.random.seed(run_seed), then the loop's draws in ITS order (one select_mb per inner
                         iteration; pnp_saga: one select_mb for the table, then select_mb + np.random.choice(hist_size, 1) per
                         iteration, algorithms/pnp_saga.py:25-29,43-47) -- so an item's trajectory equals the reference loop's
                         (and the oracle's) on the same seeds."""
    from . import engine as E
    from . import problems as P
    if problem not in PROBLEMS or algorithm not in ALGORITHMS:
        raise ValueError(f'problem in {PROBLEMS}, algorithm in {ALGORITHMS}')
    if algorithm != 'gd' and mini_batch_size is None:
        raise ValueError('mini_batch_size is required')
    if algorithm in ('svrg', 'sarah') and T2 is None:
        raise ValueError('T2 is required')
    mb, dkw = mini_batch_size, dict(denoiser_kwargs or {})

    def group_key(it):
        return None if problem == 'csmri' else it['alpha']

    def build_generator(chunk):
        a = chunk[0]['alpha']
        if problem == 'csmri':
            d = [_csmri_item_generator(images[it['image']], it, H, W) for it in chunk]
            return E.CsmriBatch(np.stack([t[0] for t in d]), np.stack([t[1] for t in d]), np.stack([t[2] for t in d]),
                                np.stack([t[3] for t in d]).reshape(len(chunk), -1), dtype=dtype)
        if problem == 'deblur':
            if deblur_scale_percent(a) != 100:
                raise ValueError('generator seeding builds scale_percent == 100 Deblur items; use seeding="legacy" for the bilinear operator')
            Bk = _minimal_kernel(H, W, kernel)
            FB = np.fft.fft(Bk)
            xs, Ys, xi = [], [], []
            for it in chunk:
                rng = np.random.default_rng(1000003 * it['seed'] + it['id'])
                x = _norm01(images[it['image']])
                Y0 = np.real(np.fft.ifft(np.fft.fft(x.ravel()) * FB)) * np.sqrt(H * W)        # DeblurSR.py:119-120
                sig = np.sqrt(np.linalg.norm(Y0) / 10 ** (it['snr'] / 10) / H / W)
                xs.append(x); Ys.append(Y0 + rng.normal(0, sig, H * W)); xi.append(rng.uniform(0.0, 1.0, H * W))
            return E.DeblurBatch(np.stack(xs), Bk, np.stack(Ys), np.stack(xi), dtype=dtype)
        M = pr_num_meas(a, H, W)
        xs, As, Ys, xi = [], [], [], []
        for it in chunk:
            rng = np.random.default_rng(1000003 * it['seed'] + it['id'])
            x = _norm01(images[it['image']])
            A = rng.standard_normal((M, H * W))
            Y0 = np.absolute(A @ x.ravel())
            sig = np.sqrt(np.linalg.norm(Y0) / 10 ** (it['snr'] / 10) / H / W)
            Y = Y0 + rng.normal(0, sig, M)
            v, lead, lead_old, prev = np.full(H * W, 2.0), 1.0, 2.0, np.ones(H * W)          # PR.py:50-63 (host: small N)
            while abs(lead - lead_old) > 1e-5 and np.linalg.norm(v - prev) > 1e-5:
                lead_old, prev = lead, v
                v = A.T @ (Y * (A @ prev)) / M
                lead = v.max()
                v = v / lead
            x0 = np.sqrt(lead) * v / np.linalg.norm(v) * np.linalg.norm(x.ravel())
            xs.append(x); As.append(A); Ys.append(Y); xi.append((x0 - x0.min()) / (x0.max() - x0.min()))
        return E.PrBatch(np.stack(xs), np.stack(As), np.stack(Ys), np.stack(xi), dtype=dtype)

    def build_legacy(chunk):
        """-> (batch, draws): per item the reference's constructor on its seed, then the loop's RNG draws on run_seed."""
        probs, draws = [], []
        n_mb = 0 if algorithm == 'gd' else n_inner + (1 if algorithm == 'saga' else 0)
        for it in chunk:
            np.random.seed(it['seed'])
            img = images[it['image']]
            if problem == 'csmri':
                p = P.CSMRI(None, H=H, W=W, sample_prob=it['alpha'], snr=it['snr'], img=img, upload=False)
            elif problem == 'deblur':
                p = P.Deblur(None, H=H, W=W, kernel=kernel, scale_percent=deblur_scale_percent(it['alpha']), snr=it['snr'], img=img,
                             dtype=dtype)
            else:
                p = P.PhaseRetrieval(None, H=H, W=W, num_meas=pr_num_meas(it['alpha'], H, W), snr=it['snr'], img=img, dtype=dtype)
            np.random.seed(run_seed)
            idx, rs = np.empty((n_mb, mb if n_mb else 0), np.int32), np.zeros(n_inner, np.int64)
            for s in range(n_mb):
                idx[s] = np.flatnonzero(p.select_mb(mb))
                if algorithm == 'saga' and s > 0:
                    rs[s - 1] = np.random.choice(hist_size, 1).item()
            probs.append(p)
            draws.append((idx, rs))
        cls = {'csmri': E.CsmriBatch, 'deblur': E.DeblurBatch, 'pr': E.PrBatch}[problem]
        return cls.from_problems(probs, dtype=dtype), draws

    class _Chunk:
        """One batch of a rank's items on its engine: built (data in HBM) by `prepare`, advanced by `advance`."""

        def __init__(self, chunk):
            self.items = chunk
            if seeding == 'legacy':
                self.batch, draws = build_legacy(chunk)
                self.idx_d = torch.from_numpy(np.stack([d[0] for d in draws], axis=1)).to(self.batch.device) if algorithm != 'gd' else None
                self.rs = np.stack([d[1] for d in draws], axis=1)                  # [n_inner][B]
            else:
                self.batch, self.idx_d, self.rs = build_generator(chunk), None, None
            kw = dict(seed=chunk[0]['id'] + 1)
            if algorithm == 'saga' and self.idx_d is not None:
                kw['idx0'] = self.idx_d[0]
            self.eng = E.make_engine(self.batch, make_prox(denoiser, **dkw), eta, T2, mb, lr_decay=lr_decay, variant=variant,
                                     algorithm=algorithm, hist_size=hist_size, **kw)
            self.done = 0

        def advance(self, n):
            eng, idx_d = self.eng, self.idx_d
            # device-drawn minibatches: whole outer iterations replay as hipGraphs (bit-identical to stepping; a rank's share
            # of a sweep is a small batch, where the ~25 launches of an inner iteration are a tenth of its time) -- when the
            # engine can be captured at all (an NLM prox ping-pongs between buffers and cannot: eager steps)
            if (graph and idx_d is None and hasattr(eng, 'run_outer') and n % T2 == 0 and eng.s % T2 == 0
                    and (n > T2 or eng.graph is not None) and eng.graph_ok()):
                eng.run_outer(n // T2)                          # (captures first when that has not happened yet: `warm`)
            else:
                for s in range(self.done, self.done + n):
                    if idx_d is None:
                        eng.step()
                    elif algorithm == 'saga':
                        eng.step(idx_d[s + 1], r=self.rs[s])
                    else:
                        eng.step(idx_d[s])
            self.done += n

        def results(self):
            tr = self.eng.psnr_trace()
            psnr0 = self.batch.psnr_init()
            z = self.eng.z.cpu().numpy()
            out = []
            for j, it in enumerate(self.items):
                r = {'id': it['id'], 'item': it, 'psnr_init': float(psnr0[j]), 'psnr_final': float(tr[-1, j]),
                     'loss': float(psnr0[j] - tr[-1, j]), 'z': z[j]}
                if problem == 'csmri':
                    r['M0'] = int(self.batch.M0[j])
                if keep_trace:
                    r['psnr_trace'] = tr[:, j].copy()
                out.append(r)
            return out

    def prepare(items):
        """Build this rank's batches (problem data resident in HBM, engines constructed): everything before the iterations."""
        groups = {}
        for it in items:
            groups.setdefault(group_key(it), []).append(it)
        return [_Chunk(g[s0:s0 + max_batch]) for g in groups.values() for s0 in range(0, len(g), max_batch)]

    def advance(state, n):
        for c in state:
            c.advance(n)

    def warm(state):
        """Capture the hipGraphs of the batches that will replay them (a timed run then starts with replays, not with the
        capture's own warm-up pass); state and results are unchanged."""
        for c in state:
            eng = c.eng
            if (graph and c.idx_d is None and hasattr(eng, 'run_outer') and eng.s % T2 == 0 and eng.graph is None and eng.graph_ok()):
                eng.capture()

    def collect(state):
        return sorted((r for c in state for r in c.results()), key=lambda r: r['id'])

    def run(items):
        state = prepare(items)
        advance(state, n_inner)
        return collect(state)

    run.prepare, run.advance, run.collect, run.warm = prepare, advance, collect, warm
    run.names = (_REF_NAMES[problem], 'CNN' if callable(denoiser) else _REF_NAMES.get(denoiser, str(denoiser)), 'pnp_' + algorithm)
    return run


def csmri_svrg_runner(images, denoiser_factory, eta, T2, mini_batch_size, n_inner, H=256, W=256, dtype=torch.float32,
                      max_batch=128, seeding='generator', algorithm='svrg', variant='svrg', run_seed=1, keep_trace=False, graph=True):
    """The config-5 runner (CSMRI + pnp_svrg, true SVRG direction): `make_runner(problem='csmri')` with a prox factory."""
    return make_runner(images, 'csmri', algorithm, denoiser_factory, eta=eta, n_inner=n_inner, mini_batch_size=mini_batch_size, T2=T2,
                       H=H, W=W, dtype=dtype, max_batch=max_batch, seeding=seeding, variant=variant, run_seed=run_seed,
                       keep_trace=keep_trace, graph=graph)


def write_csv(path, results, problem='csmri', denoiser='', algorithm='pnp_svrg', params=''):
    """CSV in the reference's schema (script_diff_sampratio_set12.py:131-136, :153-160):
    Problem,Denoiser,Algorithm,Alpha,SNR,Loss,PARAMETERS"""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Problem', 'Denoiser', 'Algorithm', 'Alpha', 'SNR', 'Loss', 'PARAMETERS'])
        for r in results:
            w.writerow([problem, denoiser, algorithm, r['item']['alpha'], r['item']['snr'], r['loss'], params])


# ---------------------------------------------------------------------------------------------- hyper-parameter search
def grid_points(grid):
    """Deterministic trial list from {'eta': [...], 'mini_batch_size': [...], 'T2': [...]}: the cartesian product in
    the key order given (the reference draws trials with hyperopt TPE, script_diff_sampratio_set12.py:116-123; a fixed
    grid is the reproducible replacement -- every item sees the same trials, so trials batch over items)."""
    import itertools
    keys = list(grid)
    return [dict(zip(keys, vals)) for vals in itertools.product(*(grid[k] for k in keys))]


def best_over_trials(per_trial):
    """per_trial: list of (params, [result dict per item]) -> one row per item with the minimum loss (ties: first trial,
    like hyperopt's best_trial on equal losses) and the parameters that achieved it.  NaN losses never win."""
    best = {}
    for params, results in per_trial:
        for r in results:
            cur = best.get(r['id'])
            loss = r['loss']
            if cur is None or (not np.isnan(loss) and (np.isnan(cur['loss']) or loss < cur['loss'])):
                best[r['id']] = {'id': r['id'], 'item': r['item'], 'loss': loss, 'params': dict(params),
                                 'psnr_init': r.get('psnr_init'), 'psnr_final': r.get('psnr_final')}
    return [best[k] for k in sorted(best)]


def grid_search(items, make_runner, grid, group=None):
    """The sweep the reference scripts run (process_img, script_diff_sampratio_set12.py:103-131): for every work item
    search the hyper-parameters and keep the best trial.  `make_runner(**params)` returns a runner as `run_sweep`
    takes; each rank runs every trial on ITS shard of the items (one batched engine per trial), the reduction over
    trials is local and the single gather at the end carries one small row per item."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    mine = shard(items, rank, world)
    per_trial = []
    for params in grid_points(grid):
        res = make_runner(**params)(mine) if mine else []
        per_trial.append((params, [{k: v for k, v in r.items() if k != 'z'} for r in res]))
    return gather_results(best_over_trials(per_trial), 0, group)


def write_tuning_csv(path, rows, problem='csmri', denoiser='', algorithm='pnp_svrg'):
    """The reference's result file, row for row (script_diff_sampratio_set12.py:131-136,153-160): a 'Results:' line,
    then Problem,Denoiser,Algorithm,Alpha,SNR,Loss,'PARAMETERS:',key,value,key,value,..."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w', newline='') as f:
        w = csv.writer(f, delimiter=',')
        w.writerow(['Results:'])
        for r in rows:
            row = [problem, denoiser, algorithm, r['item']['alpha'], r['item']['snr'], r['loss'], 'PARAMETERS:']
            for k, v in r['params'].items():
                row += [k, v]
            w.writerow(row)
