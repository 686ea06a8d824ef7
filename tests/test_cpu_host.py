"""CPU-only checks: the C-ABI library loads and exports every symbol include/pnp_hip.h declares
(no compute without a GPU), the product fails loudly without a GPU (no CPU fallback), host-side
logic (sharding, BN folding, weight import), and the world_size-2 gloo path of the sweep."""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib_path():
    from pnp_svrg_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _native.LIB_PATH


def test_header_symbols_exported():
    from pnp_svrg_amd import _native
    hdr = open(os.path.join(ROOT, 'include', 'pnp_hip.h')).read()
    declared = set(re.findall(r'\b(pnp_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 20
    h = ctypes.CDLL(_lib_path())
    missing = [n for n in declared if not hasattr(h, n)]
    assert not missing, missing
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    h.pnp_version.restype = ctypes.c_int
    assert h.pnp_version() >= 100


def test_argument_errors_are_reported_without_gpu():
    from pnp_svrg_amd import _native as N
    lib = N.lib()
    h = ctypes.c_void_p()
    assert lib.pnp_csmri_plan_create(ctypes.byref(h), 100, 100, 1, 0) == 1          # unsupported size -> PNP_ERR_ARG
    assert b'supported sizes' in lib.pnp_last_error()
    with pytest.raises(N.NativeError):
        N.call('pnp_sigma_est', None, 256, 256, 1, 0, None, None)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_no_cpu_fallback():
    import problems
    import denoisers
    from pnp_svrg_amd import _native as N
    with pytest.raises(N.NativeError):
        problems.CSMRI(os.path.join(ROOT, 'tests', 'golden', 'synth64.png'), 64, 64, 0.2, snr=20.)
    with pytest.raises(N.NativeError):
        denoisers.TVDenoiser().denoise(noisy=np.zeros((64, 64)), sigma_est=0.1)


def test_product_does_not_import_oracle():
    for d in ('pnp_svrg_amd', 'algorithms', 'problems', 'denoisers'):
        for dp, _, fs in os.walk(os.path.join(ROOT, d)):
            for f in fs:
                if f.endswith('.py'):
                    src = open(os.path.join(dp, f)).read()
                    assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), os.path.join(dp, f)


def test_dropin_import_surface():
    import algorithms, problems, denoisers
    for n in ('pnp_gd', 'pnp_sgd', 'pnp_svrg', 'pnp_saga', 'pnp_sarah', 'tune_pnp_gd', 'tune_pnp_sgd', 'tune_pnp_svrg',
              'tune_pnp_saga', 'tune_pnp_sarah'):
        assert callable(getattr(algorithms, n))
    for n in ('Problem', 'CSMRI', 'Deblur', 'PhaseRetrieval'):
        assert hasattr(problems, n)
    for n in ('Denoise', 'BM3DDenoiser', 'RealSN_DnCNNDenoiser', 'NLMDenoiser', 'TVDenoiser'):
        assert hasattr(denoisers, n)
    # flat-import style of the reference (its __init__ appends the package dir to sys.path)
    from CSMRI import CSMRI
    from TV import TVDenoiser
    from pnp_svrg import pnp_svrg
    assert CSMRI is problems.CSMRI and TVDenoiser is denoisers.TVDenoiser and pnp_svrg is algorithms.pnp_svrg
    import inspect
    sig = inspect.signature(algorithms.pnp_svrg)
    assert list(sig.parameters)[:10] == ['problem', 'denoiser', 'eta', 'tt', 'T2', 'mini_batch_size', 'verbose',
                                         'lr_decay', 'converge_check', 'diverge_check']
    assert list(inspect.signature(algorithms.pnp_saga).parameters)[:6] == ['problem', 'denoiser', 'eta', 'tt', 'mini_batch_size', 'hist_size']


def test_weight_import_and_bn_fold():
    from conftest import golden
    from pnp_svrg_amd.denoisers import dncnn_weights_from_state_dict, random_dncnn_weights
    g = dict(golden('dncnn_noise15.npz'))
    # rebuild a reference-style state dict (module. prefix, Sequential indices) and re-import it
    sd, ci = {}, 0
    for i in range(17):
        sd[f'module.dncnn.{ci}.weight'] = torch.from_numpy(g[f'conv{i}.weight'])
        if f'bn{i}.weight' in g:
            sd[f'module.dncnn.{ci + 1}.weight'] = torch.from_numpy(g[f'bn{i}.weight'])
            sd[f'module.dncnn.{ci + 1}.bias'] = torch.from_numpy(g[f'bn{i}.bias'])
            sd[f'module.dncnn.{ci + 1}.running_mean'] = torch.from_numpy(g[f'bn{i}.mean'])
            sd[f'module.dncnn.{ci + 1}.running_var'] = torch.from_numpy(g[f'bn{i}.var'])
            sd[f'module.dncnn.{ci + 1}.num_batches_tracked'] = torch.tensor(0)
            ci += 3
        else:
            ci += 2
    w = dncnn_weights_from_state_dict(sd)
    assert int(w['n_layers']) == 17 and all(np.array_equal(w[k], g[k]) for k in g)
    # RealSN layout: weight_orig / weight_u are ignored, `weight` is used
    sd2 = {'dncnn.0.weight_orig': torch.zeros(64, 1, 3, 3), 'dncnn.0.weight': torch.ones(64, 1, 3, 3),
           'dncnn.0.weight_u': torch.zeros(1, 64, 40, 40), 'dncnn.2.weight': torch.ones(1, 64, 3, 3)}
    w2 = dncnn_weights_from_state_dict(sd2)
    assert int(w2['n_layers']) == 2 and w2['conv0.weight'].sum() == 64 * 9
    r = random_dncnn_weights(17)
    assert r['conv0.weight'].shape == (64, 1, 3, 3) and r['conv16.weight'].shape == (1, 64, 3, 3) and 'bn1.var' in r


def test_shard_covers_items_once():
    from pnp_svrg_amd import sweep
    items = sweep.make_items(12, [0.1 * k for k in range(1, 11)], [20.0])
    assert len(items) == 120                                    # Set12 x 10 sampling ratios
    for world in (1, 2, 4, 8):
        parts = [sweep.shard(items, r, world) for r in range(world)]
        ids = sorted(i['id'] for p in parts for i in p)
        assert ids == list(range(120))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    out = sweep.run_sweep(items, lambda mine: [{'id': it['id'], 'v': it['alpha'] * 2} for it in mine])
    assert [r['id'] for r in out] == list(range(120))


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from pnp_svrg_amd import sweep
dist.init_process_group('gloo')
items = sweep.make_items(3, [0.1, 0.2, 0.3], [10.0, 20.0])
def runner(mine):
    return [{'id': it['id'], 'rank': dist.get_rank(), 'val': it['image'] * 100 + it['alpha'] + it['snr']} for it in mine]
res = sweep.run_sweep(items, runner)
if dist.get_rank() == 0:
    assert [r['id'] for r in res] == list(range(18)), res
    assert {r['rank'] for r in res} == {0, 1}
    assert all(r['rank'] == r['id'] % 2 for r in res)
    print('SWEEP_OK', len(res))
else:
    assert res is None
dist.barrier()
dist.destroy_process_group()
'''


def test_sweep_gloo_world2(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port), str(script), ROOT],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'SWEEP_OK 18' in out.stdout


_WORKER_GATHER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import torch.distributed as dist
from pnp_svrg_amd import sweep
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
items = sweep.make_items(5, [0.1, 0.2, 0.3], [20.0])            # 15 items on 2 ranks: 8 + 7 (ragged shards)
mine = sweep.shard(items, rank, world)
z = torch.stack([torch.full((4, 6), float(it['id'])) for it in mine])
meta = np.array([[it['id'], 100.0 + it['id']] for it in mine])
got = sweep.gather_device(z, meta, len(items))
if rank == 0:
    zz, mm = got
    assert zz.shape == (15, 4, 6) and mm.shape == (15, 2)
    assert mm[:, 0].tolist() == list(range(15)) and mm[:, 1].tolist() == [100.0 + i for i in range(15)]
    assert all(float(zz[i].min()) == float(zz[i].max()) == i for i in range(15))
    print('GATHER_OK', zz.shape[0])
else:
    assert got is None
dist.barrier()
dist.destroy_process_group()
'''


def test_gather_device_gloo_world2(tmp_path):
    """The final gather of the sweep as tensor collectives (what `bench.py --workload sweep` times inside its clock; RCCL on
    the GPU node): ragged round-robin shards of 15 items over 2 ranks arrive complete and in id order on rank 0."""
    script = tmp_path / 'worker_gather.py'
    script.write_text(_WORKER_GATHER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port), str(script), ROOT],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'GATHER_OK 15' in out.stdout
    # single process: the local rows, sorted by id
    from pnp_svrg_amd import sweep
    zz, mm = sweep.gather_device(torch.arange(6.0).reshape(3, 2, 1), np.array([[2, 0.5], [0, 0.25], [1, 0.75]]), 3)
    assert mm[:, 0].tolist() == [0, 1, 2] and zz[:, 0, 0].tolist() == [2.0, 4.0, 0.0]


def test_sweep_csv(tmp_path):
    from pnp_svrg_amd import sweep
    items = sweep.make_items(2, [0.2], [20.0])
    res = [{'id': it['id'], 'item': it, 'loss': -1.5} for it in items]
    p = tmp_path / 'out' / 'r.csv'
    sweep.write_csv(str(p), res, denoiser='DnCNN')
    rows = p.read_text().strip().splitlines()
    assert rows[0] == 'Problem,Denoiser,Algorithm,Alpha,SNR,Loss,PARAMETERS' and len(rows) == 3


def test_reference_driver_imports():
    """The import block of reference pnp_csmri.py:6-9 works against the drop-in packages."""
    ns = {}
    exec("from denoisers import *\nfrom denoisers.MMODenoise import MMODenoiser\nfrom problems import *\nfrom algorithms import *", ns)
    for n in ('CSMRI', 'Deblur', 'PhaseRetrieval', 'BM3DDenoiser', 'NLMDenoiser', 'TVDenoiser', 'RealSN_DnCNNDenoiser',
              'MMODenoiser', 'pnp_gd', 'pnp_sgd', 'pnp_svrg', 'pnp_saga', 'pnp_sarah'):
        assert n in ns, n
    # constructing the classic denoisers needs no GPU (pnp_csmri.py:18-20 does it at import time)
    ns['BM3DDenoiser'](); ns['NLMDenoiser'](); ns['TVDenoiser']()


def test_display_results_contract(tmp_path, capsys):
    """Utilities.display_results consumes the loops' result dict; the printed line reproduces the
    reference's mis-indexed format string (Utilities.py:51-53) unless asked otherwise, the CSV is indexed correctly."""
    import matplotlib
    matplotlib.use('Agg')
    import types
    from Utilities import display_results, metrics_line
    out = {'z': np.linspace(0, 1, 64), 'time_per_iter': [0.1] * 5, 'psnr_per_iter': [10.0, 11.0, 12.5, 13.0, 13.46],
           'gradient_time': 0.27, 'denoise_time': 9.75, 'algo_name': 'pnp_svrg'}
    prob = types.SimpleNamespace(H=8, W=8, color_map='gray', prob_dir=str(tmp_path) + '/')
    ax = display_results(prob, out, save_results=True)
    line = capsys.readouterr().out.strip().split('\n')[-1]
    assert line == 'Output PSNR: 13.5\tChange in PSNR: 0.27\tGradient Time: 9.75\tDenoising Time: 9.75'
    assert metrics_line(out, reference_labels=False) == 'Output PSNR: 13.5\tChange in PSNR: 3.46\tGradient Time: 0.27\tDenoising Time: 9.75'
    d = tmp_path / 'pnp_svrg'
    assert sorted(p.name for p in d.iterdir()) == ['output.csv', 'output.eps', 'psnr_over_time.eps']
    rows = (d / 'output.csv').read_text().strip().split('\n')
    assert rows[0] == 'Output PSNR,Change in PSNR,Gradient Time,Denoising Time'
    assert [float(v) for v in rows[1].split(',')] == [13.5, 3.46, 0.27, 9.75]
    assert ax.get_xlabel() == 'time (s)' and len(ax.lines) == 2


def test_grid_search_reduction_and_csv(tmp_path):
    """deterministic grid replacement for the reference's per-item hyperopt search -- trial order,
    best-trial reduction (first wins ties, NaN never wins) and the reference's CSV rows."""
    from pnp_svrg_amd import sweep
    items = sweep.make_items(2, [0.2, 0.4], [20.0])
    grid = {'eta': [1.0, 2.0, 4.0], 'T2': [5, 10]}
    pts = sweep.grid_points(grid)
    assert pts[0] == {'eta': 1.0, 'T2': 5} and pts[1] == {'eta': 1.0, 'T2': 10} and len(pts) == 6
    calls = []

    def make_runner(eta, T2):
        def run(mine):
            calls.append((eta, T2, len(mine)))
            out = []
            for it in mine:
                loss = abs(eta - 2.0) + 0.01 * T2 - it['alpha']          # minimum at eta=2, T2=5 for every item
                if it['id'] == 3 and eta == 2.0:
                    loss = float('nan')                                  # a diverged trial
                out.append({'id': it['id'], 'item': it, 'loss': loss, 'z': np.zeros(4)})
            return out
        return run
    rows = sweep.grid_search(items, make_runner, grid)
    assert [c[:2] for c in calls] == [(p['eta'], p['T2']) for p in pts] and all(c[2] == 4 for c in calls)
    assert [r['id'] for r in rows] == [0, 1, 2, 3]
    assert all(r['params'] == {'eta': 2.0, 'T2': 5} for r in rows[:3])
    assert rows[3]['params'] == {'eta': 1.0, 'T2': 5} and 'z' not in rows[0]      # first of the tied eta=1/eta=4... lowest loss, first seen
    p = tmp_path / 'hyperparam-tuning' / 'out.csv'
    sweep.write_tuning_csv(str(p), rows, denoiser='TV')
    lines = p.read_text().strip().split('\n')
    assert lines[0] == 'Results:' and len(lines) == 5
    assert lines[1].split(',')[:5] == ['csmri', 'TV', 'pnp_svrg', '0.2', '20.0']
    assert lines[1].split(',')[6:] == ['PARAMETERS:', 'eta', '2.0', 'T2', '5']


def test_w44_accumulators_untouched():
    """The MFMAs of the F(4x4,3x3) conv kernel are inline asm, so hipcc's hazard recognizer does not protect their
    results: a compiler-inserted copy or early read of an accumulator next to them would be stale in the registers the
    MFMA's last pass writes.  tools/check_w44_isa.py compiles csrc/dncnn_wino44.hip to assembly and verifies, for every
    production instantiation, that each of the 72 accumulator quads stays in its registers (16 MFMAs each) and that no
    other instruction touches an accumulator AGPR outside the epilogue, which starts with the wait states."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_w44_isa.py')], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert '0 problem(s)' in out.stdout


def test_w44b_accumulators_untouched():
    """The same for the 3 x bf16 split kernel (csrc/dncnn_wino44b.hip, conv mode 6): 18 accumulator tuples of 16 registers, 24 MFMAs
    each, untouched between a tuple's first MFMA of a region and the epilogue's wait states (tools/check_w44b_isa.py)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_w44b_isa.py')], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert '2 kernel instantiation(s) checked, 0 problem(s)' in out.stdout


def test_fused_loads_untouched():
    """The global loads of the one-kernel iteration (csrc/csmri_fused.hip) are hand-issued inline asm with hand-counted
    `s_waitcnt vmcnt(N)` in front of their uses: hipcc does not know that their results arrive later and could read, overwrite
    or SPILL a destination register behind the load (it did, under register pressure, in a first version).
    tools/check_fused_isa.py compiles the file to assembly and verifies for every instantiation, over all paths, that no
    instruction names a destination register of such a load before a wait that covers it."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_fused_isa.py')], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert out.stdout.count(' 0 violations') == 12 and 'VIOLATION' not in out.stdout      # 11 x k_svrg_iter + k_svrg_outer


def test_legacy_choice_matches_numpy():
    """`legacy_rng.choice` (C restatement of np.random.choice(..., replace=False), csrc/legacy_rng.cpp) against NumPy itself:
    the same values and dtype, and the same stream afterwards (uniform and cached-Gaussian draws that follow), for the draws the
    reference makes (problems/CSMRI.py:72 on the sampled locations, problems/problem.py:114 on range(M)) and the corner cases;
    both the in-place path and the get_state / set_state path."""
    from pnp_svrg_amd import legacy_rng
    cases = [(np.arange(3, 131073, 10), 1000), (65536, 1000), (np.arange(5), 5), (1, 1), (10, 0), (2, 1), (625, 624),
             (np.sort(np.random.RandomState(7).choice(65536, 13107, replace=False)), 13107)]
    for inplace in (None, False):
        for seed in (0, 3, 2 ** 31 - 1):
            for pool, size in cases:
                np.random.seed(seed)
                np.random.randn(3)                               # leaves a cached Gaussian and an odd position
                want = [np.random.choice(pool, size, replace=False) for _ in range(3)]
                tail_w = (np.random.rand(), np.random.randn(), np.random.get_state()[2])
                np.random.seed(seed)
                np.random.randn(3)
                legacy_rng._inplace = inplace
                got = [legacy_rng.choice(pool, size) for _ in range(3)]
                tail_g = (np.random.rand(), np.random.randn(), np.random.get_state()[2])
                for w, g in zip(want, got):
                    assert w.dtype == g.dtype and np.array_equal(w, g)
                assert tail_w == tail_g
    legacy_rng._inplace = None
    with pytest.raises(ValueError):
        legacy_rng.choice(5, 6)


def test_problem_default_device_is_current_device(monkeypatch):
    """One process per GPU: a drop-in problem built on rank r must land on the CURRENT device (torchrun ranks call
    torch.cuda.set_device(LOCAL_RANK)), not on a fixed cuda:0 -- checked with the current device mocked to 1."""
    import pnp_svrg_amd.problems as P
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 1)
    monkeypatch.setattr(P.ops, 'require_gpu', lambda: None)
    monkeypatch.setattr(P.Problem, 'to_device', lambda self, a: torch.zeros(self.H * self.W))
    img = np.arange(64.0).reshape(8, 8)
    p = P.Problem(None, 8, 8, img=img)
    assert p.device == torch.device('cuda', 1)
    assert P.Problem(None, 8, 8, img=img, device='cuda:1').device == torch.device('cuda', 1)
    with pytest.raises(Exception, match='not the current device'):
        P.Problem(None, 8, 8, img=img, device='cuda:0')
    assert P.Problem(None, 8, 8, img=img, upload=False).device.type == 'cuda'      # host-only construction: no device touched


def test_legacy_choice_delegates_outside_its_contract():
    """pools that are not 1-D integer arrays (float / 0-d / 2-D) and odd sizes go to np.random.choice itself:
    NumPy's dtype, values, errors and stream position."""
    from pnp_svrg_amd import legacy_rng
    for pool, size in ((np.array([0.5, 1.5, 2.5, 3.5]), 3), (np.array(7), 3), (6, np.int64(2)), (np.arange(5, dtype=np.uint8), 4)):
        np.random.seed(11)
        a = legacy_rng.choice(pool, size)
        ua = np.random.uniform()
        np.random.seed(11)
        b = np.random.choice(pool, size, replace=False)
        ub = np.random.uniform()
        assert np.array_equal(a, b) and ua == ub
        assert a.dtype == b.dtype or np.asarray(pool).dtype.kind in 'iu'        # (integer pools: int64, NumPy's default int)
    with pytest.raises(ValueError):
        legacy_rng.choice(np.zeros((2, 2), int), 1)
    with pytest.raises(ValueError):
        legacy_rng.choice(3, 5)
