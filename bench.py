#!/usr/bin/env python3
"""bench.py -- PnP-SVRG inner-iterations/s on 256x256 CSMRI (20 % sampling) + DnCNN-17 prox.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload dncnn|tv|saga-nlm]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one inner iteration of pnp_svrg (reference algorithms/pnp_svrg.py:41-94: minibatch SVRG direction via
the masked-FFT gradient, step, estimate_sigma, DnCNN prox, PSNR error sum) for a batch of B independent
reconstructions per GPU, including the outer full-gradient refresh every T2 = 10 steps.  Inputs (problems, network
weights) are resident in HBM before the timed region; minibatches are drawn on the device (one launch per outer
iteration).  Masks are Bernoulli like the reference's (problems/CSMRI.py:43-45), so every problem has its own M0.
Every rank runs its own B problems (weak scaling; the only collective is the final gather of results, after the
timed region).

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel = the 64->64 3x3
conv on the f32 matrix cores; `frac` = EXECUTED MFMA FLOP/s over the f32 matrix peak), `cpu_baseline` (the oracle,
i.e. the CPU port of the reference path, timed on this box's host cores on a bounded sample) and -- at N = 1 in the
default configuration -- `secondary`: BASELINE configs 2 (TV prox, HBM-bound) and 4 (Deblur + NLM prox + pnp_saga)
timed in the same run with their own `roofline` / `cpu_baseline`.

`--gpus N` without torchrun (WORLD_SIZE unset) launches the N ranks itself (torch.distributed.run as a child process,
before this process touches the GPU) and exits with its status.
"""
import argparse
import json
import os
import subprocess
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
# the reference's blur has gain 1/sqrt(N) (B = kernel / N, fft_blur x sqrt(N)): gradients are O(1e-9) and a minibatch's
# Lipschitz constant ~ 5e-9; with the reference's table initialisation (all 50 rows = the gradient at the U(0,1) start,
# pnp_saga.py:29-31) the direction keeps that bias for ~hist steps, hence the small step size
SAGA_ETA, SAGA_MB, SAGA_HIST, SAGA_SNR = 5e6, 3000, 50, 20.0
FLOP_MID_PER_IMAGE = 2 * 9 * 64 * 64 * H * W          # one 64->64 3x3 conv layer, direct form
WINOGRAD_REDUCTION = {'5': 4.0, '1': 1.5, '0': 1.0,   # F(4x4,3x3) executes 1/4 of the direct form's multiply-adds, F(2,3) along x 2/3;
                      '6': 4.0 / 6.0}                  # the 3 x bf16 split form: 1/4 of the multiply-adds, six bf16 products for each
BF16_MFMA_PEAK_TFLOPS = 2500.0                        # same guide: dense bf16 peak
F32_MFMA_PEAK_TFLOPS = 157.3                          # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
F32_VALU_PEAK_TFLOPS = 157.3                          # same guide: vector f32 FMA peak (NLM prox)
HBM_PEAK_GBS = 8000.0
TV_BYTES_PER_ITER = 2368 * 1024                       # algorithmic bytes of one config-2 problem-iteration
NLM_FLOP_PER_PIXEL = 121 * 25 * 5                     # 11x11 window x 5x5 patch x (sub, mul, sub, mul, add), before early exits
SAGA_BYTES_PER_ITER = 8 * H * W * 4                   # table update: read g, old slot, prev, sum, z; write slot, sum, z


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=None,
                    help='independent reconstructions per GPU (default: 120 = one Set12 x 10 sampling-ratio sweep for dncnn, '
                         '1024 for tv (SURVEY 8d batch list), 64 for saga-nlm)')
    ap.add_argument('--workload', default='dncnn', choices=['dncnn', 'tv', 'saga-nlm', 'sweep'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the config-2 / config-4 secondary measurements')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='gloo + PNP_BENCH_ONE_DEVICE=1 rehearses the N>1 path with all ranks on GPU 0')
    ap.add_argument('--graph', action='store_true',
                    help='replay one outer iteration (T2 steps) per hipGraph launch; --steps/--warmup are rounded up to multiples of T2')
    ap.add_argument('--conv', default=None, choices=['f32-winograd44', 'f32-winograd', 'f32-direct', 'bf16x3-winograd44'],
                    help='conv kernel of the DnCNN prox (default: f32-winograd44 = F(4x4,3x3), or PNP_DNCNN_WINOGRAD; f32-winograd = F(2,3) along x, '
                         'f32-direct = implicit GEMM, bit for bit an fmaf chain; bf16x3-winograd44 = opt-in F(4x4,3x3) on the bf16 matrix cores '
                         'with exact three-way splits: fp32-class accuracy, not the reference\'s arithmetic)')
    ap.add_argument('--host-minibatches', action='store_true', help='pre-draw minibatch index lists on the host')
    ap.add_argument('--no-outer-kernel', action='store_true', help='A/B: config 2 as one launch per inner iteration instead of one per outer iteration')
    ap.add_argument('--no-fold', action='store_true', help='A/B: the outer full-gradient refresh as launches of its own instead of '
                                                            'folded into the first inner iteration (one-kernel iteration only)')
    ap.add_argument('--cpu-baseline-child', default=None, metavar='WORKLOAD:SECONDS', help=argparse.SUPPRESS)
    return ap.parse_args()


def _synth_image(rng):
    x = rng.random((H, W))
    p5 = np.pad(x, 2, mode='wrap')
    return sum(p5[i:i + H, j:j + W] for i in range(5) for j in range(5)) / 25.0


def cpu_baseline(workload, weights, budget_s=12.0):
    """The CPU baseline in a child process that never touches the GPU, so that the oracle's NumPy / torch-CPU threads do not
    share a process with the HIP runtime's.  (On the shared hosts of the GPU pool the DnCNN figure still moves between 6 and 31
    inner-iterations/s from run to run with the placement of its 16 threads, tools/check_child_baseline.py.)  `weights` is
    ignored here: the child loads the same fixture."""
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-child', f'{workload}:{budget_s}']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=60 + 20 * budget_s)
    if out.returncode != 0:
        raise RuntimeError('cpu_baseline child failed: ' + out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def _cpu_baseline_here(workload, weights, budget_s=12.0):
    """The oracle (oracle/: NumPy + torch-CPU restatement of the reference path) on the host cores:
    the same inner iteration, one problem at a time, for ~budget_s seconds."""
    from oracle import problems as op, denoise as od
    # a one-GPU box gets a 16-CPU share: more threads than that only oversubscribe the cgroup
    threads = min(torch.get_num_threads(), len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    np.random.seed(0)
    img = _synth_image(np.random.default_rng(0))
    if workload == 'saga-nlm':
        p = op.Deblur(None, H=H, W=W, kernel='Minimal', scale_percent=100, snr=SAGA_SNR, img=img)
        d = od.NLMDenoiser()
        d.sigma = 1.0
        z = np.copy(p.Xinit)
        g0 = p.grad_stoch(z, p.select_mb(SAGA_MB)) / SAGA_MB
        table, prev = [g0] * SAGA_HIST, g0
        n, t0 = 0, time.perf_counter()
        while True:
            mb = p.select_mb(SAGA_MB)
            r = np.random.choice(SAGA_HIST, 1).item()
            table[r] = p.grad_stoch(z, mb) / SAGA_MB
            v = table[r] - prev + sum(table) / SAGA_HIST
            prev = table[r]
            z = z - SAGA_ETA * v
            z0 = z.reshape(H, W)
            z0 = d.denoise(noisy=z0, sigma_est=od.estimate_sigma(z0))
            p.PSNR(z0)
            z = z0.ravel()
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s and n >= 2:
                break
        what = f'{H}x{W} Deblur ("Minimal" kernel) + NLM prox, pnp_saga hist {SAGA_HIST}'
    else:
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
        what = f'{H}x{W} CSMRI + {workload} prox'
    return {'value': n / el, 'unit': 'inner-iters/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n} inner iterations of 1 problem ({what}, oracle/ NumPy' + ('+torch-CPU fp32 net' if workload == 'dncnn' else '') + f') in {el:.1f} s'}


def _traffic(key):
    """(HBM bytes per launch / step from the builder's PMC profile, where that number comes from) -- a constant read from
    profiles/traffic.json, NOT something the run that prints the line measured."""
    tj = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tj):
        try:
            t = json.load(open(tj))
            return t.get(key), t.get(key + '_source')
        except Exception:
            return None, None
    return None, None


class Workload:
    """One configuration: builds the batch + engine, runs timed steps (device time from HIP events recorded on the
    stream the kernels are launched on), produces its roofline object."""

    def __init__(self, name, B, rank, a, weights=None):
        from pnp_svrg_amd.engine import CsmriBatch, DeblurBatch, make_engine, DnCNNProx, TVProx, NLMProx
        self.name, self.B, self.a = name, B, a
        if name == 'saga-nlm':
            self.batch = DeblurBatch.synthetic(B, H, W, 'Minimal', SAGA_SNR, seed=100 + rank)
            self.prox = NLMProx()
            self.eng = make_engine(self.batch, self.prox, SAGA_ETA, T2, SAGA_MB, algorithm='saga', hist_size=SAGA_HIST, seed=1 + rank)
            self.mbsize = SAGA_MB
        else:
            self.batch = CsmriBatch.synthetic(B, H, W, SAMPLE_PROB, SNR, seed=100 + rank)
            self.prox = DnCNNProx(weights, NET_SIGMA) if name == 'dncnn' else TVProx()
            self.eng = make_engine(self.batch, self.prox, ETA, T2, MB, variant='svrg', seed=1 + rank,
                                   fold_outer=not getattr(a, 'no_fold', False))
            self.mbsize = MB
        self.n_draw = 0
        self.idx = None
        self.done = 0

    def predraw(self, n):
        self.n_draw = min(n, 64)
        self.idx = self.batch.draw_minibatches(self.n_draw, self.mbsize, seed=7)

    def run(self, n):
        eng = self.eng
        # config 2: whole outer iterations as ONE launch each (the workgroup that owns a problem runs its T2 inner iterations back
        # to back, pnp_csmri_svrg_outer_iteration) whenever the request is for whole outer iterations
        if (self.idx is None and not getattr(self.a, 'no_outer_kernel', False) and hasattr(eng, 'outer_kernel_ok') and eng.outer_kernel_ok()
                and n % T2 == 0 and eng.s % T2 == 0):
            eng.run_outer(n // T2)
            self.done += n
            return
        for _ in range(n):
            if self.idx is not None:
                eng.step(self.idx[self.done % self.n_draw])
            else:
                eng.step()
            self.done += 1

    def roofline(self, dt_step):
        a, B = self.a, self.B
        if self.name == 'dncnn':
            ms, launches = self.prox.plan.profile_end()
            flops = FLOP_MID_PER_IMAGE * B
            alg = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            mode = os.environ.get('PNP_DNCNN_WINOGRAD', '5')
            red = WINOGRAD_REDUCTION.get(mode, 4.0)
            ex = alg / red
            peak = BF16_MFMA_PEAK_TFLOPS if mode == '6' else F32_MFMA_PEAK_TFLOPS
            return {'bound': 'mfma',
                    'kernel': {'6': 'pnp::w44b::k_mid_wino44b (64->64 3x3 conv, Winograd F(4x4,3x3) on v_mfma_f32_32x32x16_bf16: every fp32 factor '
                                    'split exactly into three bf16 terms, six products each; opt-in, fp32-class accuracy)',
                               '5': 'pnp::w44::k_mid_wino44 (64->64 3x3 conv, Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32)',
                               '1': 'pnp::k_mid_wino (64->64 3x3 conv, Winograd F(2,3) along x on v_mfma_f32_16x16x4_f32)',
                               '0': 'pnp::k_mid (64->64 3x3 conv, direct implicit GEMM on v_mfma_f32_16x16x4_f32)'}.get(mode, mode),
                    # achieved / frac: what the matrix cores EXECUTE per second against their peak (mode 6: bf16 FLOPs against the bf16 peak)
                    'achieved': round(ex, 2), 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': round(ex / peak, 4), 'traffic': _traffic(f'k_mid_B{B}')[0] if mode != '6' else None,
                    'traffic_source': _traffic(f'k_mid_B{B}')[1] if mode != '6' else None,
                    'launch_ms': round(ms, 4), 'launches_timed': launches,
                    'flops_per_launch': int(flops / red), 'algorithmic_flops_per_launch': flops,
                    'algorithmic_tflops': round(alg, 2), 'winograd_reduction': red}
        if self.name == 'tv':
            # no single dominant kernel (rows_inv / prox / cols / rows_fwd ~ 20-27 % each): the whole step against HBM
            alg = TV_BYTES_PER_ITER * B
            ach = alg / dt_step / 1e9
            return {'bound': 'hbm',
                    'kernel': 'whole inner iteration = 1/T2 of pnp::k_svrg_outer (one launch per outer iteration: the workgroup that owns a problem '
                              'runs the folded full-gradient refresh and its T2 inner iterations back to back; each iteration = SVRG step through the '
                              'masked FFT, noise estimate, Haar prox, error sum, image register-resident, every global access 16 bytes per lane) '
                              '+ 1/T2 of k_draw_thr; with --no-outer-kernel: one pnp::k_svrg_iter launch per inner iteration',
                    'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                    'traffic': _traffic(f'tv_step_B{B}')[0], 'traffic_source': _traffic(f'tv_step_B{B}')[1], 'bytes_per_step': alg}
        # saga-nlm: the NLM prox dominates (VALU-bound patch search); the table update is the HBM-bound part
        ms = self.nlm_ms
        fl = NLM_FLOP_PER_PIXEL * H * W * B
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {'bound': 'valu',
                'kernel': 'pnp::k_nlm_strip<float,5,5> (11x11 search window x 5x5 patches, neighbour patches in a register strip fed from an LDS-staged tile, f32; nominal FLOPs before '
                          'the reference\'s early exits and border clipping)',
                'achieved': round(ach, 2), 'peak': F32_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / F32_VALU_PEAK_TFLOPS, 4),
                'frac_kind': 'NOMINAL: 5 FLOPs per patch element before the reference\'s early exits, over the f32 vector peak -- not a hardware fraction',
                'valu_issue_frac': 0.53,
                'valu_issue_frac_source': 'profiles/r02e_nlm_sq_counters.txt (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of k_nlm_strip, builder PMC run): '
                                          'the share of wave time in which the vector ALU issues -- the hardware-side figure next to the nominal frac',
                'traffic': None, 'launch_ms': round(ms, 4), 'flops_per_launch': fl,
                'saga_table_update': {'bound': 'hbm', 'bytes_per_step': SAGA_BYTES_PER_ITER * B,
                                      'launch_ms': round(self.saga_ms, 4),
                                      'achieved': round(SAGA_BYTES_PER_ITER * B / (self.saga_ms * 1e-3) / 1e9, 1) if self.saga_ms > 0 else None,
                                      'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                      'frac': round(SAGA_BYTES_PER_ITER * B / (self.saga_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if self.saga_ms > 0 else None}}

    def time_nlm_and_update(self, reps=5):
        """HIP-event timing of the two config-4 kernels in isolation (torch events on the launch stream)."""
        from pnp_svrg_amd import ops
        eng, b = self.eng, self.batch
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        sse = torch.empty(b.B, dtype=torch.float64, device=b.device)
        buf = torch.empty_like(eng.z)
        ops.nlm2d(eng.z, sigma_in=self.prox.sig, xrec=b.xrec, out=buf, sse=sse)
        torch.cuda.synchronize()
        e[0].record()
        for _ in range(reps):
            ops.nlm2d(eng.z, sigma_in=self.prox.sig, xrec=b.xrec, out=buf, sse=sse)
        e[1].record()
        zc, ts = eng.z.clone(), eng.tsum.clone()
        e[2].record()
        for _ in range(reps):
            ops.saga_table_update(zc, eng.g, eng.table[1], eng.table[0], ts, 0.0, 1.0 / eng.hist)
        e[3].record()
        torch.cuda.synchronize()
        self.nlm_ms = e[0].elapsed_time(e[1]) / reps
        self.saga_ms = e[2].elapsed_time(e[3]) / reps


SWEEP_IMAGES, SWEEP_ALPHAS = 12, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]


class SweepWorkload:
    """BASELINE config 5: Set12-shaped batch x sampling-ratio sweep (12 images x 10 ratios = 120 independent work items, reference
    script_diff_sampratio_set12.py:113-146), DnCNN prox, pnp_svrg, dealt round-robin over the ranks by `sweep.shard` (STRONG
    scaling: the 120 items are the whole job whatever N), every rank's items one mixed-ratio batch on the engine, and ONE
    collective: the final gather of the reconstructions and PSNRs -- inside the timed region."""

    def __init__(self, rank, world, weights, n_images=SWEEP_IMAGES, alphas=SWEEP_ALPHAS):
        from pnp_svrg_amd import sweep
        from pnp_svrg_amd.engine import DnCNNProx
        self.sweep, self.name = sweep, 'sweep'
        rng = np.random.default_rng(2024)
        imgs = [_synth_image(rng) for _ in range(n_images)]
        self.items = sweep.make_items(n_images, alphas, [SNR])
        self.runner = sweep.make_runner(imgs, 'csmri', 'svrg', lambda: DnCNNProx(weights, NET_SIGMA), eta=ETA, n_inner=T2,
                                        mini_batch_size=MB, T2=T2, H=H, W=W, seeding='generator', max_batch=128)
        self.mine = sweep.shard(self.items, rank, world)
        self.state = self.runner.prepare(self.mine)              # problem data resident in HBM, engines built
        self.runner.warm(self.state)                             # hipGraph of an outer iteration captured (before any timing)
        self.B = len(self.mine)

    def run(self, n):
        self.runner.advance(self.state, n)

    def gather(self):
        """the final gather (z and PSNRs of every item to rank 0): RCCL when the process group is NCCL"""
        zs, meta = [], []
        for c in self.state:
            tr = c.eng.sse_log[(c.eng.n_prox - 1) % c.eng.n_log]         # squared errors of the last iteration, on the device
            zs.append(c.eng.z)
            meta.append(torch.stack([torch.tensor([it['id'] for it in c.items], dtype=torch.float64, device=tr.device), tr], 1))
        z = torch.cat(zs) if zs else torch.zeros((0, H, W), dtype=torch.float32, device='cuda')
        m = torch.cat(meta) if meta else torch.zeros((0, 2), dtype=torch.float64, device='cuda')
        return self.sweep.gather_device(z, m, len(self.items))


def measure(w, steps, warmup, sync_all, graph=False):
    """warmup untimed steps, then exactly `steps` timed ones bracketed by sync (+ barrier); returns seconds."""
    if graph:
        w.eng.capture()
        w.eng.run_outer(warmup // T2)
    else:
        w.run(warmup)
    sync_all()
    if w.name == 'dncnn' and not graph:
        w.prox.plan.profile_begin(steps + 8)
    t0 = time.perf_counter()
    if graph:
        w.eng.run_outer(steps // T2)
    else:
        w.run(steps)
    sync_all()
    return time.perf_counter() - t0


def run_sweep_bench(rank, world, weights, steps, warmup, sync_all, dist, cdev):
    """Time config 5: `steps` inner iterations of all 120 items + the final gather.  Returns (seconds (max over ranks), warmup
    steps actually run, the workload, gathered (z, meta) on rank 0)."""
    w = SweepWorkload(rank, world, weights)
    if steps % T2 == 0:
        warmup = -(-max(warmup, 1) // T2) * T2                 # whole outer iterations: the timed ones then replay as hipGraphs
    w.run(warmup)
    w.gather()                                                  # (communicator / buffers warm)
    sync_all()
    t0 = time.perf_counter()
    w.run(steps)
    got = w.gather()
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, warmup, w, got


def sweep_fields(dt, steps, w, got, world):
    """value / config / roofline of a sweep measurement (rank 0)."""
    n_items = len(w.items)
    z, meta = got
    psnr = 10 * np.log10(1.0 / (meta[:, 1].numpy() / (H * W)))
    flops = FLOP_MID_PER_IMAGE * 15 * n_items / WINOGRAD_REDUCTION['5']     # executed by the 15 middle layers per step, whole job
    ex = flops * steps / dt / 1e12 / world                                   # per GPU
    return {'value': round(n_items * steps / dt, 2), 'unit': 'item-inner-iters/s', 'ms_per_step': round(dt / steps * 1e3, 4),
            'scaling': 'strong',
            'config': {'workload': f'{SWEEP_IMAGES} images x {len(SWEEP_ALPHAS)} sampling ratios = {n_items} work items (reference '
                                   f'script_diff_sampratio_set12.py), {H}x{W} CSMRI Bernoulli masks, pnp_svrg (true SVRG direction, T2={T2}, '
                                   f'mb={MB}), DnCNN-17 prox; items dealt round-robin over ranks, each rank ONE mixed-ratio batch; final gather of '
                                   'z and PSNR to rank 0 INSIDE the timed region',
                       'items_total': n_items, 'items_per_gpu': -(-n_items // world),
                       'parallelism': f'items sharded x{world}; one gather (RCCL) at the end'},
            'roofline': {'bound': 'mfma', 'kernel': 'whole step (gradient kernels, 17-layer network, gather) priced as the F(4x4,3x3) conv layers\' executed FLOPs',
                         'achieved': round(ex, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ex / F32_MFMA_PEAK_TFLOPS, 4),
                         'traffic': None, 'note': 'per GPU; includes everything in the timed region, so it is a lower bound of the conv kernel\'s own fraction'},
            'psnr_db': {'after_timed_steps_mean': float(np.mean(psnr)), 'items_gathered': int(z.shape[0])}}


def metric_name(workload):
    return {'dncnn': 'PnP-SVRG inner-iters/sec, 256x256 CSMRI+DnCNN', 'tv': 'PnP-SVRG inner-iters/sec, 256x256 CSMRI+TV',
            'saga-nlm': 'PnP-SAGA iters/sec, 256x256 Deblur+NLM',
            'sweep': 'PnP-SVRG item-inner-iters/sec, Set12-shaped batch x 10 sampling ratios, 256x256 CSMRI+DnCNN (strong scaling)'}[workload]


def workload_desc(workload, wdesc):
    if workload == 'saga-nlm':
        return (f'pnp_saga (hist_size={SAGA_HIST}, mb={SAGA_MB}) on {H}x{W} Deblur ("Minimal" kernel, scale 100 %), NLM prox '
                f'(patch 5x5, search 11x11)')
    return (f'pnp_svrg (true SVRG direction, T2={T2}, mb={MB}) on {H}x{W} CSMRI, {int(SAMPLE_PROB * 100)}% Bernoulli mask, '
            + (f'DnCNN-17 prox ({wdesc})' if workload == 'dncnn' else 'TV (Haar BayesShrink) prox'))


def _weights():
    """the reference's own DnCNN sigma=15 weights (committed fixture) when present, else random init"""
    wfile = os.path.join(ROOT, 'tests', 'golden', 'dncnn_noise15.npz')
    if os.path.exists(wfile):
        return dict(np.load(wfile)), 'reference DnCNN_noise15 weights'
    from pnp_svrg_amd.denoisers import random_dncnn_weights
    return random_dncnn_weights(17, seed=0), 'random-init weights'


def main():
    a = parse()
    if a.cpu_baseline_child is not None:                        # (no GPU call on this path)
        wl, budget = a.cpu_baseline_child.split(':')
        print(json.dumps(_cpu_baseline_here(wl, _weights()[0] if wl == 'dncnn' else None, float(budget))))
        return
    world_env = os.environ.get('WORLD_SIZE')
    if a.gpus > 1 and world_env is None:
        # launched the documented way without torchrun: start the N ranks as children (this process has not touched
        # the GPU) and hand their status back
        import socket
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f'[bench] --gpus {a.gpus} without WORLD_SIZE: launching {" ".join(cmd)}', file=sys.stderr)
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', '0'))
    world = int(world_env or '1')
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
    if a.gpus != world and rank == 0:
        print(f'[bench] --gpus {a.gpus} but WORLD_SIZE {world}: using WORLD_SIZE', file=sys.stderr)

    if a.conv is not None:
        os.environ['PNP_DNCNN_WINOGRAD'] = {'f32-winograd44': '5', 'f32-winograd': '1', 'f32-direct': '0', 'bf16x3-winograd44': '6'}[a.conv]
    from pnp_svrg_amd import ops
    ops.require_gpu()

    B = a.batch if a.batch is not None else {'dncnn': 120, 'tv': 1024, 'saga-nlm': 64, 'sweep': 120}[a.workload]
    weights, wdesc = _weights()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    cdev = 'cuda' if (dist is None or a.backend == 'nccl') else 'cpu'      # where collective buffers live
    if a.workload == 'sweep':
        dt, warm, w, got = run_sweep_bench(rank, world, weights, a.steps, a.warmup, sync_all, dist, cdev)
        if rank == 0:
            f = sweep_fields(dt, a.steps, w, got, world)
            line = {'metric': metric_name('sweep'), 'value': f['value'], 'unit': f['unit'], 'n_gpus': world, 'steps': a.steps, 'warmup': warm,
                    'ms_per_step': f['ms_per_step'], 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
                    'data': 'synthetic', 'config': f['config'], 'roofline': f['roofline'],
                    'cpu_baseline': None if a.no_cpu_baseline else cpu_baseline('dncnn', weights), 'psnr_db': f['psnr_db']}
            print(json.dumps(line))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    w = Workload(a.workload, B, rank, a, weights)
    if a.host_minibatches:
        w.predraw(a.steps + a.warmup)
    if a.workload == 'tv' and not a.host_minibatches and a.steps % T2 == 0:
        a.warmup = -(-max(a.warmup, 1) // T2) * T2               # whole outer iterations: the timed ones then run one launch each
    if a.graph:
        assert w.idx is None and a.workload != 'saga-nlm', '--graph needs device-side minibatch draws and the SVRG engine'
        a.steps = -(-a.steps // T2) * T2
        a.warmup = -(-max(a.warmup, 1) // T2) * T2
    dt = measure(w, a.steps, a.warmup, sync_all, a.graph)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if a.workload == 'saga-nlm':
        w.time_nlm_and_update()
    roofline = None if (a.graph and a.workload == 'dncnn') else w.roofline(dt / a.steps)

    # final gather of the results (the only collective on this path; outside the timed region)
    trace = w.eng.psnr_trace()
    psnr0 = float(np.mean(w.batch.psnr_init()))
    final_psnr = torch.from_numpy(np.ascontiguousarray(trace[-1])).to(cdev)
    if dist is not None:
        gathered = [torch.empty_like(final_psnr) for _ in range(world)] if rank == 0 else None
        dist.gather(final_psnr, gathered, dst=0)
        all_psnr = torch.stack(gathered).cpu().numpy() if rank == 0 else None
    else:
        all_psnr = final_psnr.cpu().numpy()[None]

    if rank == 0:
        secondary = None
        if world == 1 and a.workload == 'dncnn' and not a.no_secondary and not a.graph and a.conv is None and a.batch is None:
            # BASELINE configs 2 and 4 in the same run, each with its own roofline (and a short CPU baseline)
            secondary = {}
            del w.eng, w.prox, w.batch
            torch.cuda.empty_cache()
            for name, bb, st in (('tv', 1024, 100), ('saga-nlm', 64, 20)):
                w2 = Workload(name, bb, rank, a)
                dt2 = measure(w2, st, 10 if name == 'tv' else 3, sync_all)
                if name == 'saga-nlm':
                    w2.time_nlm_and_update()
                tr2 = w2.eng.psnr_trace()
                secondary[name] = {'metric': metric_name(name), 'value': round(bb * st / dt2, 2), 'unit': 'inner-iters/s',
                                   'steps': st, 'ms_per_step': round(dt2 / st * 1e3, 4), 'dtype': 'f32',
                                   'config': {'workload': workload_desc(name, ''), 'batch_per_gpu': bb},
                                   'roofline': w2.roofline(dt2 / st),
                                   'cpu_baseline': None if a.no_cpu_baseline else cpu_baseline(name, None, budget_s=6.0),
                                   'psnr_db': {'initial_mean': float(np.mean(w2.batch.psnr_init())),
                                               'after_timed_steps_mean': float(np.mean(tr2[-1]))}}
                del w2
                torch.cuda.empty_cache()
            # config 3 once more with the opt-in conv mode 6 (3 x bf16 split F(4x4,3x3); fp32-class accuracy, not the reference's
            # arithmetic operation for operation -- which is why `value` stays on the exact-fp32 kernel)
            os.environ['PNP_DNCNN_WINOGRAD'] = '6'
            try:
                w6 = Workload('dncnn', 120, rank, a, weights)
                dt6 = measure(w6, 10, 2, sync_all)
                tr6 = w6.eng.psnr_trace()
                secondary['dncnn-bf16x3'] = {'metric': metric_name('dncnn'), 'value': round(120 * 10 / dt6, 2), 'unit': 'inner-iters/s',
                                             'steps': 10, 'ms_per_step': round(dt6 / 10 * 1e3, 4), 'dtype': 'bf16x3 (three-way exact split of fp32, fp32 accumulation)',
                                             'config': {'workload': workload_desc('dncnn', wdesc) + ', conv mode 6 (bf16x3-winograd44)', 'batch_per_gpu': 120},
                                             'note': 'opt-in kernel, fp32-class accuracy (tests: test_bf16x3_conv_mode); currently SLOWER than the exact-fp32 '
                                                     'default that `value` is measured on -- DESIGN 3.1',
                                             'roofline': w6.roofline(dt6 / 10),
                                             'psnr_db': {'initial_mean': float(np.mean(w6.batch.psnr_init())),
                                                         'after_timed_steps_mean': float(np.mean(tr6[-1]))}}
                del w6
            finally:
                del os.environ['PNP_DNCNN_WINOGRAD']
            torch.cuda.empty_cache()
            # config 5 on one GPU: the sweep driver with its gather inside the clock (the N > 1 runs are `--workload sweep --gpus N`)
            dt5, warm5, w5, got5 = run_sweep_bench(rank, world, weights, 20, 10, sync_all, None, cdev)
            f5 = sweep_fields(dt5, 20, w5, got5, 1)
            secondary['sweep'] = {'metric': metric_name('sweep'), 'value': f5['value'], 'unit': f5['unit'], 'steps': 20, 'warmup': warm5,
                                  'ms_per_step': f5['ms_per_step'], 'dtype': 'f32', 'scaling': 'strong', 'config': f5['config'],
                                  'roofline': f5['roofline'], 'psnr_db': f5['psnr_db']}
            del w5, got5
            torch.cuda.empty_cache()
        cpu = None
        if not a.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(a.workload, weights)
        value = world * B * a.steps / dt
        line = {
            'metric': metric_name(a.workload),
            'value': round(value, 2), 'unit': 'inner-iters/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(dt / a.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': workload_desc(a.workload, wdesc),
                       'batch_per_gpu': B, 'problems_total': world * B, 'parallelism': f'replicas x{world} (no data-path collective)'},
            'roofline': roofline, 'cpu_baseline': cpu,
            'psnr_db': {'initial_mean': psnr0, 'after_timed_steps_mean': float(np.mean(all_psnr))},
        }
        if secondary is not None:
            line['secondary'] = secondary
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
