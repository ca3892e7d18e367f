"""Shared helpers for the test-suite (no product code here)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def manifest():
    with open(os.path.join(GOLDEN, 'ref_manifest.json')) as f:
        return json.load(f)


def rel_l2(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def weights(npz, prefix='w::', requires_grad=False, device='cpu'):
    sd = {}
    for k in npz.files:
        if k.startswith(prefix):
            t = torch.from_numpy(npz[k]).to(device)
            if requires_grad and t.is_floating_point():
                t.requires_grad_(True)
            sd[k[len(prefix):]] = t
    return sd


def noise_seq(npz, tag):
    n = int(npz[f'{tag}_n_noise'])
    return [torch.from_numpy(npz[f'{tag}_noise_{i}']) for i in range(n)]


def randomise(model, gen, scale=0.05):
    """tests/golden/make_ref_golden.py::randomise: perturb every parameter with draws of a CPU generator (reproducible), so a
    fixture can store ONE base weight set and name per-seed variants by their generator seed."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('freqs'):
                continue
            p.add_(scale * torch.randn(p.shape, generator=gen))


def _hash_values(n, seed):
    """n reproducible values in [-0.5, 0.5) from an integer hash (exact in fp64 on every host: no random generator, no libm)."""
    i = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B1)
    i = (i ^ (i >> np.uint64(15))) * np.uint64(0x2C1B3C6D) & np.uint64(0xFFFFFFFF)
    i = (i ^ (i >> np.uint64(12))) * np.uint64(0x297A2D39) & np.uint64(0xFFFFFFFF)
    i = i ^ (i >> np.uint64(15))
    return i.astype(np.float64) / 4294967296.0 - 0.5


def guidance_input(tshape, shape, ori, seed):
    """Inputs of the smoke control objective (smoke/inference_2d.py:30-66) for tests/golden/ref_guidance.npz, regenerated identically by the
    fixture generator and by the tests: the state x [B, F, 42, H, W] in network units (coefficients inside `shape`, zero padding outside, a
    full smoke-out channel), RESCALER [1, 1, 42, 1, 1] and the initial density init_u [B, H_ori, W_ori] (not rescaled)."""
    b, f, c, h, w = tshape
    tc, hc, wc = shape
    x = np.zeros(tshape, dtype=np.float64)
    x[:, :tc, :, :hc, :wc] = 0.6 * _hash_values(b * tc * c * hc * wc, seed).reshape(b, tc, c, hc, wc)
    x[:, :tc, -1] = 0.6 * _hash_values(b * tc * h * w, seed + 1).reshape(b, tc, h, w)
    resc = np.linspace(1.0, 9.0, c).reshape(1, 1, c, 1, 1)
    init_u = 2.0 * _hash_values(b * ori[1] * ori[2], seed + 2).reshape(b, ori[1], ori[2])
    return torch.from_numpy(x).float(), torch.from_numpy(resc).float(), torch.from_numpy(init_u).float()
