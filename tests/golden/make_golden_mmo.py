#!/usr/bin/python3
"""Golden vectors for the MMO denoiser (reference denoisers/MMODenoise.py), produced by RUNNING the reference's own
`simple_CNN` network class and `MMODenoiser.denoise` wrapper (torch CPU fp32) on seeded weights.

The reference's MMO checkpoints (denoisers/checkpoints/pretrained/*.pth) are whole pickled modules and are NOT loaded
(only weights_only=True loads are allowed on reference files, and those refuse them).  The weights here are a seeded
He-normal draw (so that 20 LeakyReLU layers keep O(1) activations), written out as plain arrays.

`denoisers/cnn/cnn.py` imports torchvision (absent in this image) for its training dataset class only; an empty
stand-in module satisfies the import, nothing from it is called.

    /usr/bin/python3 tests/golden/make_golden_mmo.py
"""
import importlib.util
import os
import sys
import types
import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules.setdefault('torchvision', types.ModuleType('torchvision'))
sys.path.insert(0, REF + '/denoisers')                       # MMODenoise.py falls back to `from denoiser import ...`
spec = importlib.util.spec_from_file_location('ref_mmo', REF + '/denoisers/MMODenoise.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

torch.manual_seed(0)
torch.set_num_threads(8)
net = ref.simple_CNN(n_ch_in=1, n_ch_out=1, n_ch=64, nl_type='relu', depth=20, bn=False)
with torch.no_grad():
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            fan_in = m.in_channels * 9
            m.weight.normal_(0.0, (2.0 / fan_in) ** 0.5)
            m.bias.uniform_(-0.1, 0.1)
    net.out_conv.weight.mul_(0.1)
net.eval()
den = ref.MMODenoiser(model=net, channels=1, cuda=False, sigma=0.01)

from PIL import Image                                          # noqa: E402
img = np.asarray(Image.open(os.path.join(HERE, 'synth256.png')), dtype=np.float64) / 255.0
rng = np.random.default_rng(7)
out = {}
for name, (h, w) in (('sq', (64, 64)), ('rect', (64, 96))):
    x = img[16:16 + h, 32:32 + w]
    x = (x - x.min()) / (x.max() - x.min())
    noisy = 1.3 * x - 0.15 + 0.05 * rng.standard_normal((h, w))        # some pixels < 0 and > 1: both clamps act
    y = den.denoise(noisy)
    assert y.shape == (h, w) and y.dtype == np.float32
    out[f'{name}_in'] = noisy
    out[f'{name}_out'] = y
    print(name, 'in range', noisy.min(), noisy.max(), 'out range', y.min(), y.max(),
          'clipped lo/hi', int((y == 0).sum()), int((y == 1).sum()), 'mean |y - clip(in)|', np.abs(y - np.clip(noisy, 0, 1)).mean())
assert den.t == 2
sd = net.state_dict()
names = ['in_conv'] + [f'conv_list.{i}' for i in range(18)] + ['out_conv']
out['n_layers'] = np.int64(20)
out['negative_slope'] = np.float64(torch.nn.LeakyReLU().negative_slope)
for i, n in enumerate(names):
    out[f'conv{i}.weight'] = sd[n + '.weight'].numpy()
    out[f'conv{i}.bias'] = sd[n + '.bias'].numpy()
np.savez_compressed(os.path.join(HERE, 'mmo_seeded.npz'), **out)
print({k: getattr(v, 'shape', v) for k, v in out.items() if not k.startswith('conv')})
