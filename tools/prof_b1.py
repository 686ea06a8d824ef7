"""B = 1 inner iterations of the drop-in pnp_svrg + TV loop, eagerly, for a rocprofv3 --kernel-trace --stats run."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import problems, denoisers
from pnp_svrg_amd.algorithms import _SvrgGraph
np.random.seed(0)
p = problems.CSMRI(os.path.join(ROOT, 'tests', 'golden', 'synth256.png'), H=256, W=256, sample_prob=0.2, snr=20., dtype=torch.float32)
run = _SvrgGraph(p, denoisers.TVDenoiser(), 2e3, 10, 1000, 'svrg', 4096)
run.upload([[p._select_mb_locs(1000) for _ in range(10)]])
run.idx.copy_(run.all_idx[0].reshape(10, 1, 1000))
for _ in range(30):
    run.outer_body()
torch.cuda.synchronize()
