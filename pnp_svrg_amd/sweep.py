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
    """Static round-robin: item i -> rank i % world (items of equal cost; no exchange afterwards)."""
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


def csmri_svrg_runner(images, denoiser_factory, eta, T2, mini_batch_size, n_inner, H=256, W=256, dtype=torch.float32,
                      max_batch=64):
    """Default runner: CSMRI + pnp_svrg (true SVRG direction) on the batched engine.
    images: list of HxW arrays; items with the same alpha are batched together (equal M0 per batch)."""
    from .engine import CsmriBatch, make_engine

    def run(items):
        results = []
        by_alpha = {}
        for it in items:
            by_alpha.setdefault(it['alpha'], []).append(it)
        for alpha, group in by_alpha.items():
            for s0 in range(0, len(group), max_batch):
                chunk = group[s0:s0 + max_batch]
                xs, masks, Ys, xinits = [], [], [], []
                m0 = int(round(alpha * H * W))
                for it in chunk:
                    rng = np.random.default_rng(1000003 * it['seed'] + it['id'])
                    x = np.asarray(images[it['image']], np.float64)
                    x = (x - x.min()) / (x.max() - x.min())
                    mk = np.zeros(H * W, np.uint8)
                    mk[rng.choice(H * W, m0, replace=False)] = 1
                    mk = mk.reshape(H, W)
                    Y0 = mk * np.fft.fft2(x)
                    sig = np.sqrt(np.linalg.norm(Y0.ravel()) / 10 ** (it['snr'] / 10) / H / W)
                    Y = Y0 + mk * rng.normal(0, sig, (H, W))
                    xi = np.absolute(np.fft.ifft2(Y))
                    xs.append(x); masks.append(mk); Ys.append(Y); xinits.append((xi - xi.min()) / (xi.max() - xi.min()))
                batch = CsmriBatch(np.stack(xs), np.stack(masks), np.stack(Ys), np.stack(xinits).reshape(len(chunk), -1), dtype=dtype)
                eng = make_engine(batch, denoiser_factory(), eta, T2, mini_batch_size, variant='svrg')
                idx = batch.draw_minibatches(n_inner, mini_batch_size, seed=chunk[0]['id'] + 1)
                for s in range(n_inner):
                    eng.step(idx[s])
                tr = eng.psnr_trace()
                psnr0 = np.around(10 * np.log10(1.0 / ((batch.xinit - batch.xrec) ** 2).reshape(len(chunk), -1).mean(1).double().cpu().numpy()), 2)
                z = eng.z.cpu().numpy()
                for j, it in enumerate(chunk):
                    results.append({'id': it['id'], 'item': it, 'psnr_init': float(psnr0[j]), 'psnr_final': float(tr[-1, j]),
                                    'loss': float(psnr0[j] - tr[-1, j]), 'z': z[j]})
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
