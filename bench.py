#!/usr/bin/env python3
"""bench.py -- PnP-SVRG inner-iterations/s on 256x256 CSMRI (20 % sampling) + DnCNN-17 prox.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload dncnn|tv]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one inner iteration of pnp_svrg (reference algorithms/pnp_svrg.py:41-94: minibatch
SVRG direction via the masked-FFT gradient, step, estimate_sigma, DnCNN prox, PSNR error sum) for a
batch of B independent reconstructions per GPU, including the outer full-gradient refresh every
T2 = 10 steps.  Inputs (problems, network weights) are resident in HBM before the timed region; minibatches are drawn
on the device inside each step.  Every rank runs its own B problems (weak scaling; the only
collective is the final gather of results, after the timed region).

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel =
the 64->64 3x3 conv on the f32 matrix cores) and `cpu_baseline` (the oracle, i.e. the CPU port of
the reference path, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 256
SAMPLE_PROB, SNR = 0.2, 20.0
ETA, T2, MB = 2e3, 10, 1000
NET_SIGMA = 15
FLOP_MID_PER_IMAGE = 2 * 9 * 64 * 64 * H * W          # one 64->64 3x3 conv layer
F32_MFMA_PEAK_TFLOPS = 157.3                          # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0                         # same guide: ~2.5 PF dense bf16/f16 (only for --conv f16x3)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=120,
                    help='independent reconstructions per GPU (default 120 = one Set12 x 10 sampling-ratio sweep, BASELINE config 5)')
    ap.add_argument('--workload', default='dncnn', choices=['dncnn', 'tv'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo + PNP_BENCH_ONE_DEVICE=1 rehearses the N>1 path with all ranks on GPU 0')
    ap.add_argument('--graph', action='store_true',
                    help='replay one outer iteration (T2 steps) per hipGraph launch; --steps/--warmup are rounded up to multiples of T2')
    ap.add_argument('--conv', default=None, choices=['f32-winograd', 'f32-direct', 'f16x3'],
                    help='conv kernel of the DnCNN prox (default: f32-winograd, or PNP_DNCNN_WINOGRAD).  f16x3 = opt-in '
                         'split-fp16 products with fp32 accumulation (fp32-class accuracy, not the reference arithmetic): '
                         'the line then says dtype "f32 via 3 x f16 split" and prices the conv against the f16 matrix peak')
    ap.add_argument('--fused-tv', action='store_true',
                    help='tv workload: pnp_csmri_grad_prox_tv (one kernel for step + noise estimate + prox; measured slower)')
    ap.add_argument('--host-minibatches', action='store_true', help='pre-draw minibatch index lists on the host')
    return ap.parse_args()


def cpu_baseline(workload, weights, budget_s=12.0):
    """The oracle (oracle/: NumPy + torch-CPU restatement of the reference path) on the host cores:
    the same inner iteration, one problem at a time, for ~budget_s seconds."""
    from oracle import problems as op, denoise as od
    # a one-GPU box gets a 16-CPU share: more threads than that only oversubscribe the cgroup
    threads = min(torch.get_num_threads(), len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    np.random.seed(0)
    rng = np.random.default_rng(0)
    x = rng.random((H, W))
    p5 = np.pad(x, 2, mode='wrap')
    img = sum(p5[i:i + H, j:j + W] for i in range(5) for j in range(5)) / 25.0
    p = op.CSMRI(None, H=H, W=W, sample_prob=SAMPLE_PROB, snr=SNR, img=img)
    d = od.DnCNNDenoiser(weights, NET_SIGMA) if workload == 'dncnn' else od.TVDenoiser()
    z = np.copy(p.Xinit)
    mu = p.grad_full(z)
    w = np.copy(z)
    n, t0 = 0, time.perf_counter()
    while True:
        if n % T2 == 0:
            mu = p.grad_full(z)
            w = np.copy(z)
        mb = p.select_mb(MB)
        v = (p.grad_stoch(z, mb) - p.grad_stoch(w, mb)) / MB + mu
        z = z - ETA * v
        z0 = z.reshape(H, W)
        z0 = d.denoise(noisy=z0, sigma_est=od.estimate_sigma(z0))
        p.PSNR(z0)
        z = z0.ravel()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s and n >= 3:
            break
    return {'value': n / el, 'unit': 'inner-iters/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n} inner iterations of 1 problem ({H}x{W} CSMRI + {workload} prox, oracle/ NumPy+torch-CPU fp32 net) in {el:.1f} s'}


def main():
    a = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.environ.get('PNP_BENCH_ONE_DEVICE') == '1':
            local = 0
        torch.cuda.set_device(local)
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo')
    else:
        dist = None
        torch.cuda.set_device(0)
    if a.gpus != world and rank == 0 and world > 1:
        print(f'[bench] --gpus {a.gpus} but WORLD_SIZE {world}: using WORLD_SIZE', file=sys.stderr)

    if a.conv is not None:
        os.environ['PNP_DNCNN_WINOGRAD'] = {'f32-winograd': '1', 'f32-direct': '0', 'f16x3': '3'}[a.conv]
    from pnp_svrg_amd import ops
    from pnp_svrg_amd.engine import CsmriBatch, make_engine, DnCNNProx, TVProx
    from pnp_svrg_amd.denoisers import random_dncnn_weights
    ops.require_gpu()

    B = a.batch
    # the reference's own DnCNN sigma=15 weights (committed fixture) when present, else random init
    wfile = os.path.join(ROOT, 'tests', 'golden', 'dncnn_noise15.npz')
    weights = dict(np.load(wfile)) if os.path.exists(wfile) else random_dncnn_weights(17, seed=0)
    wdesc = 'reference DnCNN_noise15 weights' if os.path.exists(wfile) else 'random-init weights'
    batch = CsmriBatch.synthetic(B, H, W, SAMPLE_PROB, SNR, seed=100 + rank)
    prox = DnCNNProx(weights, NET_SIGMA) if a.workload == 'dncnn' else TVProx()
    eng = make_engine(batch, prox, ETA, T2, MB, variant='svrg', seed=1 + rank, fused=bool(a.fused_tv and a.workload == 'tv'))
    # minibatches are drawn ON THE DEVICE inside every step (pnp_csmri_draw_minibatch), like the reference
    # draws them inside its timed gradient phase (pnp_svrg.py:52); --host-minibatches pre-draws index lists
    n_draw = min(a.steps + a.warmup, 64)
    idx = batch.draw_minibatches(n_draw, MB, seed=1 + rank) if a.host_minibatches else None

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if a.graph:
        assert idx is None, '--graph needs device-side minibatch draws'
        a.steps = -(-a.steps // T2) * T2
        a.warmup = -(-max(a.warmup, 1) // T2) * T2
        eng.capture()
        eng.run_outer(a.warmup // T2)
    else:
        for s in range(a.warmup):
            eng.step(idx[s % n_draw] if idx is not None else None)
    sync_all()
    if a.workload == 'dncnn' and not a.graph:
        prox.plan.profile_begin(a.steps + 8)
    t0 = time.perf_counter()
    if a.graph:
        eng.run_outer(a.steps // T2)
    else:
        for s in range(a.steps):
            eng.step(idx[(a.warmup + s) % n_draw] if idx is not None else None)
    sync_all()
    dt = time.perf_counter() - t0
    cdev = 'cuda' if (dist is None or a.backend == 'nccl') else 'cpu'      # where collective buffers live
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roofline = None
    if a.workload == 'dncnn' and not a.graph:
        ms, launches = prox.plan.profile_end()
        flops = FLOP_MID_PER_IMAGE * B
        ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic = None
        tj = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get(f'k_mid_B{B}')
            except Exception:
                traffic = None
        mode = os.environ.get('PNP_DNCNN_WINOGRAD', '1')
        wino = mode not in ('0', '3')
        if mode == '3':
            executed = 3.0 * ach                                # three fp16 MFMAs per product
            roofline = {'bound': 'mfma',
                        'kernel': 'pnp::k_mid_f16x3 (64->64 3x3 conv, every fp32 operand split into two fp16 terms, three '
                                  'v_mfma_f32_16x16x32_f16 per product, fp32 accumulation; OPT-IN, not the reference arithmetic)',
                        'achieved': round(executed, 2), 'peak': F16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(executed / F16_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                        'launch_ms': round(ms, 4), 'launches_timed': launches, 'flops_per_launch': 3 * flops,
                        'algorithmic_tflops': round(ach, 2)}
        else:
          roofline = {'bound': 'mfma',
                    'kernel': ('pnp::k_mid_wino (64->64 3x3 conv, Winograd F(2,3) along x on v_mfma_f32_16x16x4_f32: 2/3 of the '
                               'direct form\'s multiply-adds, so the ALGORITHMIC rate can exceed the matrix-core peak)') if wino
                              else 'pnp::k_mid (64->64 3x3 conv, direct implicit GEMM on v_mfma_f32_16x16x4_f32)',
                    'achieved': round(ach, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / F32_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic,
                    'launch_ms': round(ms, 4), 'launches_timed': launches,
                    'flops_per_launch': flops,
                    # what the matrix cores actually execute (Winograd F(2,3) needs 2/3 of the multiply-adds)
                    'executed_tflops': round(ach * (2.0 / 3.0 if wino else 1.0), 2),
                    'executed_frac_of_mfma_peak': round(ach * (2.0 / 3.0 if wino else 1.0) / F32_MFMA_PEAK_TFLOPS, 4)}

    if a.workload == 'tv':
        # no single dominant kernel (rows_inv / prox / cols / rows_fwd ~ 20-27 % each): the whole step against HBM
        alg = 2368 * 1024 * B                                    # 2 368 KiB per problem-iteration
        ach = alg / (dt / a.steps) / 1e9
        kern = ('whole inner iteration (k_draw_mb + k_rows_fwd + k_cols + ' +
                ('k_rows_inv_prox: step, noise estimate, prox and error fused)' if a.fused_tv else 'k_rows_inv + k_prox_tv)'))
        traffic = None
        tj = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tj) and not a.fused_tv:
            try:
                traffic = json.load(open(tj)).get(f'tv_step_B{B}')     # PMC bytes of one whole step (all five kernels)
            except Exception:
                traffic = None
        roofline = {'bound': 'hbm', 'kernel': kern,
                    'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                    'traffic': traffic, 'bytes_per_step': alg}

    # final gather of the results (the only collective on this path; outside the timed region)
    trace = eng.psnr_trace()
    psnr0 = float(np.mean(np.around(10 * np.log10(1.0 / ((batch.xinit - batch.xrec) ** 2).reshape(B, -1).mean(1).cpu().numpy()), 2)))
    final_psnr = torch.from_numpy(np.ascontiguousarray(trace[-1])).to(cdev)
    if dist is not None:
        gathered = [torch.empty_like(final_psnr) for _ in range(world)] if rank == 0 else None
        dist.gather(final_psnr, gathered, dst=0)
        all_psnr = torch.stack(gathered).cpu().numpy() if rank == 0 else None
    else:
        all_psnr = final_psnr.cpu().numpy()[None]

    if rank == 0:
        cpu = None
        if not a.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(a.workload, weights)
        value = world * B * a.steps / dt
        line = {
            'metric': 'PnP-SVRG inner-iters/sec, 256x256 CSMRI+DnCNN' if a.workload == 'dncnn'
                      else 'PnP-SVRG inner-iters/sec, 256x256 CSMRI+TV',
            'value': round(value, 2), 'unit': 'inner-iters/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32 via 3 x f16 split products (opt-in)' if (a.workload == 'dncnn' and os.environ.get('PNP_DNCNN_WINOGRAD') == '3') else 'f32',
            'data': 'synthetic',
            'config': {'workload': f'pnp_svrg (true SVRG direction, T2={T2}, mb={MB}) on {H}x{W} CSMRI, '
                                   f'{int(SAMPLE_PROB * 100)}% mask, '
                                   + (f'DnCNN-17 prox ({wdesc})' if a.workload == 'dncnn' else 'TV (Haar BayesShrink) prox'),
                       'batch_per_gpu': B, 'problems_total': world * B, 'parallelism': f'replicas x{world} (no data-path collective)'},
            'roofline': roofline, 'cpu_baseline': cpu,
            'psnr_db': {'initial_mean': psnr0, 'after_timed_steps_mean': float(np.mean(all_psnr))},
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
