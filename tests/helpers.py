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
