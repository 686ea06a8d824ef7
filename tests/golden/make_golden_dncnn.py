#!/usr/bin/python3
"""Golden vectors for the DnCNN prox, produced by RUNNING the reference network class
(/root/reference/denoisers/DeepDenoisers/model/models.py, imported by file path) on the
reference's own weights (DnCNN_noise{5,15,40}.pth, loaded with weights_only=True), torch
CPU fp32.  The RealSN_DnCNN_noise*.pth files are absent from the reference checkout
(SURVEY F10); DnCNN_noise*.pth has the identical inference graph.

    /usr/bin/python3 tests/golden/make_golden_dncnn.py

Writes weights as plain arrays (data, not code) + inputs/outputs.
"""
import importlib.util
import os
import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)
torch.set_num_threads(8)

spec = importlib.util.spec_from_file_location('ref_models', REF + '/denoisers/DeepDenoisers/model/models.py')
ref_models = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_models)


def load(sigma):
    net = ref_models.DnCNN(channels=1, num_of_layers=17)
    sd = torch.load(f'{REF}/denoisers/DeepDenoisers/Pretrained_models/DnCNN_noise{sigma}.pth',
                    map_location='cpu', weights_only=True)
    sd = {k[len('module.'):] if k.startswith('module.') else k: v for k, v in sd.items()}
    missing = net.load_state_dict(sd, strict=True)
    print(sigma, missing)
    net.eval()
    return net, sd


def export(sd):
    """state dict -> flat arrays: conv{i}.weight, bn{i}.{weight,bias,mean,var}; i = conv ordinal."""
    out = {'n_layers': np.int64(17)}
    conv_idx = sorted({int(k.split('.')[1]) for k in sd if k.endswith('.weight') and sd[k].dim() == 4})
    for i, ci in enumerate(conv_idx):
        out[f'conv{i}.weight'] = sd[f'dncnn.{ci}.weight'].numpy().astype(np.float32)
        bi = ci + 1
        if f'dncnn.{bi}.running_mean' in sd:
            out[f'bn{i}.weight'] = sd[f'dncnn.{bi}.weight'].numpy()
            out[f'bn{i}.bias'] = sd[f'dncnn.{bi}.bias'].numpy()
            out[f'bn{i}.mean'] = sd[f'dncnn.{bi}.running_mean'].numpy()
            out[f'bn{i}.var'] = sd[f'dncnn.{bi}.running_var'].numpy()
    return out


def wrapper(net, noisy, sigma):
    """reference denoisers/RealSN_DnCNN.py:16-42 with the hard-coded .cuda() dropped
    (no GPU in the build container): the same arithmetic around the CPU net."""
    m, n = noisy.shape
    xt = np.copy(noisy)
    lo, hi = np.min(xt), np.max(xt)
    xt = (xt - lo) / (hi - lo)
    sr = 1.0 + sigma / 255.0 / 2.0
    ss = (1 - sr) / 2.0
    xt = xt * sr + ss
    with torch.no_grad():
        r = net(torch.from_numpy(np.reshape(xt, (1, 1, m, n))).type(torch.FloatTensor)).numpy()
    x = xt - np.reshape(r, (m, n))
    x = (x - ss) / sr
    return x * (hi - lo) + lo


def main():
    den = np.load(os.path.join(HERE, 'denoise.npz'))
    z256, z64 = den['r256_z0'], den['s64_z0']
    out = {}
    for sigma in (5, 15, 40):
        net, sd = load(sigma)
        if sigma == 15:
            np.savez_compressed(os.path.join(HERE, 'dncnn_noise15.npz'), **export(sd))
        else:
            # other noise levels: keep only a checksum of the weights + outputs on the small case
            out[f'w{sigma}_checksum'] = np.array([float(sum(v.double().abs().sum() for v in sd.values()))])
        out[f'den64_s{sigma}'] = wrapper(net, z64, sigma) if sigma == 15 else np.zeros(0)
    net, sd = load(15)
    out['den256_s15'] = wrapper(net, z256, 15)
    # raw network I/O + per-layer activations on 64^2 for kernel bring-up
    x64 = ((z64 - z64.min()) / (z64.max() - z64.min())).astype(np.float32)
    with torch.no_grad():
        t = torch.from_numpy(x64)[None, None]
        acts = {}
        for li, layer in enumerate(net.dncnn):
            t = layer(t)
            acts[li] = t
    out['net64_in'] = x64
    out['net64_out'] = acts[len(net.dncnn) - 1][0, 0].numpy()
    out['net64_act_relu0'] = acts[1][0].numpy()            # after conv0+ReLU          (64,64,64)
    out['net64_act_relu1'] = acts[4][0].numpy()            # after conv1+BN+ReLU
    out['net64_act_relu15'] = acts[len(net.dncnn) - 2][0].numpy()   # input of the last conv
    x256 = ((z256 - z256.min()) / (z256.max() - z256.min())).astype(np.float32)
    with torch.no_grad():
        out['net256_in'] = x256
        out['net256_out'] = net(torch.from_numpy(x256)[None, None])[0, 0].numpy()
    np.savez_compressed(os.path.join(HERE, 'dncnn_io.npz'), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
