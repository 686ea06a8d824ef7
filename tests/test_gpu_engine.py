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
