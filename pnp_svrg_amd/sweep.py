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


def _csmri_item_generator(img, it, H, W):
    """One work item's data from a Generator stream keyed by the item (fast path of the sweeps): Bernoulli mask like
    the reference (problems/CSMRI.py:43-45), masked spectrum + real noise on the support (:29-33), |ifft2| init."""
    rng = np.random.default_rng(1000003 * it['seed'] + it['id'])
    x = np.asarray(img, np.float64)
    x = (x - x.min()) / (x.max() - x.min())
    mk = (rng.random((H, W)) < it['alpha']).astype(np.uint8)
    Y0 = mk * np.fft.fft2(x)
    sig = np.sqrt(np.linalg.norm(Y0.ravel()) / 10 ** (it['snr'] / 10) / H / W)
    Y = Y0 + mk * rng.normal(0, sig, (H, W))
    xi = np.absolute(np.fft.ifft2(Y))
    return x, mk, Y, ((xi - xi.min()) / (xi.max() - xi.min())).ravel()


def csmri_svrg_runner(images, denoiser_factory, eta, T2, mini_batch_size, n_inner, H=256, W=256, dtype=torch.float32,
                      max_batch=128, seeding='generator', algorithm='svrg', variant='svrg', run_seed=1, keep_trace=False, graph=True):
    """Default runner: CSMRI + pnp_svrg (true SVRG direction) on the batched engine.
    images: list of HxW arrays.  ALL of a rank's items -- any mix of sampling ratios, i.e. masks with different
    numbers of sampled points -- go through the engine together (chunks of max_batch): per-problem 1/M0 and
    per-problem minibatch thresholds make the batch independent of alpha.
    seeding='generator': per-item data from a Generator stream, minibatches drawn on the device;
    seeding='legacy'   : per item exactly what the reference does -- np.random.seed(item seed), the CSMRI constructor's
                         draws in its order (mask, noise; problems/CSMRI.py:12-41), then np.random.seed(run_seed) and one
                         select_mb draw per inner iteration (algorithms/pnp_svrg.py:52) from the legacy stream -- so an
                         item's trajectory equals the reference loop's (and the oracle's) on the same seeds.
    graph=False steps eagerly where whole outer iterations would otherwise replay as hipGraphs (same results)."""
    from .engine import CsmriBatch, make_engine
    from . import problems as P

    def run(items):
        results = []
        for s0 in range(0, len(items), max_batch):
            chunk = items[s0:s0 + max_batch]
            xs, masks, Ys, xinits, idx = [], [], [], [], None
            if seeding == 'legacy':
                idx = np.empty((n_inner, len(chunk), mini_batch_size), np.int32)
                for j, it in enumerate(chunk):
                    np.random.seed(it['seed'])
                    p = P.CSMRI(None, H=H, W=W, sample_prob=it['alpha'], snr=it['snr'], img=images[it['image']], upload=False)
                    xs.append(p.Xrec); masks.append(p.mask); Ys.append(p.Y); xinits.append(p.Xinit)
                    np.random.seed(run_seed)
                    for s in range(n_inner):
                        idx[s, j] = np.flatnonzero(p.select_mb(mini_batch_size))
            else:
                for it in chunk:
                    x, mk, Y, xi = _csmri_item_generator(images[it['image']], it, H, W)
                    xs.append(x); masks.append(mk); Ys.append(Y); xinits.append(xi)
            batch = CsmriBatch(np.stack(xs), np.stack(masks), np.stack(Ys), np.stack(xinits).reshape(len(chunk), -1), dtype=dtype)
            eng = make_engine(batch, denoiser_factory(), eta, T2, mini_batch_size, variant=variant, algorithm=algorithm,
                              seed=chunk[0]['id'] + 1)
            idx_d = torch.from_numpy(idx).to(batch.device) if idx is not None else None
            # device-drawn minibatches: whole outer iterations replay as hipGraphs (bit-identical to stepping; a rank's share
            # of a sweep is a small batch, where the ~25 launches of an inner iteration are a tenth of its time)
            if (graph and idx_d is None and hasattr(eng, 'run_outer') and n_inner % T2 == 0 and n_inner > T2
                    and getattr(eng, 'lr_decay', 1.0) == 1.0 and getattr(eng.prox, 'denoise_strength', 0.0) == 0.0):
                eng.run_outer(n_inner // T2)
            else:
                for s in range(n_inner):
                    eng.step(idx_d[s]) if idx_d is not None else eng.step()
            tr = eng.psnr_trace()
            psnr0 = batch.psnr_init()
            z = eng.z.cpu().numpy()
            for j, it in enumerate(chunk):
                r = {'id': it['id'], 'item': it, 'psnr_init': float(psnr0[j]), 'psnr_final': float(tr[-1, j]),
                     'loss': float(psnr0[j] - tr[-1, j]), 'z': z[j], 'M0': int(batch.M0[j])}
                if keep_trace:
                    r['psnr_trace'] = tr[:, j].copy()
                results.append(r)
        return sorted(results, key=lambda r: r['id'])
    return run


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
