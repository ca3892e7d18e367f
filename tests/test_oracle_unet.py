"""Pins oracle/unet_ref.py against outputs of the imported reference (tests/golden/ref_unet*.npz)."""
import numpy as np
import pytest
import torch

from oracle import unet_ref as U
from tests.helpers import load_npz, manifest, rel_l2, weights

M = manifest()


def _run(name, fwd, cfg):
    g = load_npz(name)
    sd = weights(g, requires_grad=True)
    x = torch.from_numpy(g['x']).requires_grad_(True)
    out = fwd(sd, x, torch.from_numpy(g['t']), **cfg)
    assert rel_l2(out, g['out']) < 2e-6
    (out * torch.from_numpy(g['gout'])).sum().backward()
    assert rel_l2(x.grad, g['gx']) < 5e-6
    for k in g.files:
        if k.startswith('gn::'):
            key = k[4:]
            ref = float(g[k])
            got = sd[key].grad.double().norm().item()
            assert abs(got - ref) <= 2e-5 * max(ref, 1e-6), (key, got, ref)
        if k.startswith('g::'):
            assert rel_l2(sd[k[3:]].grad, g[k]) < 2e-5, k


def test_unet3d_tiny():
    c = M['unet3d_tiny']
    _run('ref_unet3d_tiny.npz', U.unet3d_forward, dict(dim=c['dim'], dim_mults=tuple(c['dim_mults']), groups=c['resnet_groups']))


def test_unet2d_tiny():
    c = M['unet2d_tiny']
    _run('ref_unet2d_tiny.npz', U.unet2d_forward, dict(dim=c['dim'], dim_mults=tuple(c['dim_mults']), groups=c['resnet_block_groups']))


def test_unet2d_grouped():
    c = M['unet2d_g4']
    _run('ref_unet2d_g4.npz', U.unet2d_forward, dict(dim=c['dim'], dim_mults=tuple(c['dim_mults']), groups=c['resnet_block_groups']))


def test_relative_position_buckets():
    pos = torch.arange(24)
    b = U.relative_position_bucket(pos[None, :] - pos[:, None])
    assert b.shape == (24, 24) and int(b.min()) == 0 and int(b.max()) <= 31
    assert int(b[0, 0]) == 0 and int(b[5, 0]) == 5 and int(b[0, 5]) == 16 + 5      # sign convention: n = q - k
    assert int(b[23, 0]) == 8 + int(np.floor(np.log(23 / 8) / np.log(4) * 8))
