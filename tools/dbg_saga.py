import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pnp_svrg_amd.engine import DeblurBatch, NLMProx, make_engine
b = DeblurBatch.synthetic(64, 256, 256, 'Minimal', 20.0, seed=100)
e = make_engine(b, NLMProx(), 5e7, 10, 3000, algorithm='saga', hist_size=50, seed=1)
print('psnr0', b.psnr_init()[:4], 'g absmax', float(e.g.abs().max()), 'nan g', bool(torch.isnan(e.g).any()))
for s in range(24):
    e.step()
    tr = e.psnr_trace()[-1]
    print(s, 'nan psnr', int(np.isnan(tr).sum()), tr[:3], 'sig', e.prox.sig.cpu().numpy()[:3], 'nan sig', int(torch.isnan(e.prox.sig).sum()),
          'nan z', int(torch.isnan(e.z).sum()), 'zmax', float(torch.nan_to_num(e.z).abs().max()), 'nan g', int(torch.isnan(e.g).sum()), 'nan sum', int(torch.isnan(e.tsum).sum()))
