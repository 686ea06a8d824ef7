#!/usr/bin/python3
"""Golden vectors for the 4-layer SimpleCNN / RealSN_SimpleCNN denoisers (reference
denoisers/DeepDenoisers/utils/utils.py:16-25 -> model/SimpleCNN_models.py), produced by RUNNING the reference's
own network class on its own checkpoints (torch CPU fp32, weights_only=True).

    /usr/bin/python3 tests/golden/make_golden_simplecnn.py
"""
import io
import contextlib
import os
import sys
import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF + '/denoisers/DeepDenoisers')          # the reference file imports `model.conv_sn_chen`
from model.SimpleCNN_models import DnCNN                        # noqa: E402

den = np.load(os.path.join(HERE, 'denoise.npz'))
z64 = den['s64_z0']
x64 = ((z64 - z64.min()) / (z64.max() - z64.min())).astype(np.float32)
out = {'net64_in': x64}
for name, lip in (('SimpleCNN', 0.0), ('RealSN_SimpleCNN', 1.0)):
    with contextlib.redirect_stdout(io.StringIO()):
        net = DnCNN(1, num_of_layers=4, lip=lip, no_bn=True)
    sd = torch.load(f'{REF}/denoisers/DeepDenoisers/Pretrained_models/{name}_noise15.pth', map_location='cpu', weights_only=True)
    print(name, net.load_state_dict(sd, strict=True))
    net.eval()
    with torch.no_grad():
        out[f'{name}_out'] = net(torch.from_numpy(x64)[None, None])[0, 0].numpy()
    # the weights the inference graph uses (`weight`; RealSN files also hold weight_orig / weight_u)
    for i, ci in enumerate((0, 2, 4, 6)):
        out[f'{name}_conv{i}.weight'] = sd[f'dncnn.{ci}.weight'].numpy()
np.savez_compressed(os.path.join(HERE, 'simplecnn_noise15.npz'), **out)
print({k: v.shape for k, v in out.items()})
