#!/usr/bin/python3
"""DnCNN prox at the other two noise levels the reference ships (sigma = 5, 40): the reference network class on the
reference's own DnCNN_noise{5,40}.pth (weights_only=True), torch CPU fp32 -- weights as plain arrays + wrapper outputs
at 64 x 64 and 256 x 256 (same inputs as dncnn_io.npz).

    /usr/bin/python3 tests/golden/make_golden_dncnn_r2.py
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_dncnn import load, export, wrapper     # noqa: E402  (the round-1 harness: reference class by file path)


def main():
    den = np.load(os.path.join(HERE, 'denoise.npz'))
    z256, z64 = den['r256_z0'], den['s64_z0']
    for sigma in (5, 40):
        net, sd = load(sigma)
        out = export(sd)
        out['den64'] = wrapper(net, z64, sigma)
        out['den256'] = wrapper(net, z256, sigma)
        np.savez_compressed(os.path.join(HERE, f'dncnn_noise{sigma}.npz'), **out)
        print(sigma, out['den64'].shape, out['den256'].shape)


if __name__ == '__main__':
    main()
