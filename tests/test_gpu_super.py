"""Super-resolution variants of the diffusion operators (rows D4-D7 with `low` conditioning, BASELINE configs[4]) against the
reference (tests/golden/make_ref_super_golden.py): smoke space-SR (control) and time-SR models, Burgers SR model."""
import json
import os
import sys

import pytest
import torch

from tests.helpers import GOLDEN, load_npz, rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL, CHAIN_TOL = 1e-5, 1e-5
DDIM3_TOL = 1e-5      # measured 8.0e-6 and 4.7e-6


@pytest.fixture(scope='module')
def trees():
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion as GD1
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion as GD2
    return dict(Unet2D=Unet2D, GD1=GD1, Unet3D=Unet3D_with_Conv3D, GD2=GD2)


G = load_npz('ref_super_diffusion.npz')
with open(os.path.join(GOLDEN, 'ref_super_manifest.json')) as f:
    META = json.load(f)


def sub(tag):
    pre = tag + '::'
    return {k[len(pre):]: G[k] for k in G.files if k.startswith(pre)}


def check_grad_norms(module, gz):
    params = dict(module.named_parameters())
    n = 0
    for k, v in gz.items():
        if k.startswith('gn::'):
            ref = float(v); got = params[k[4:]].grad.double().norm().item()
            assert abs(got - ref) <= 2e-4 * max(ref, 1e-7), (k, got, ref)
            n += 1
    assert n > 20


@pytest.mark.parametrize('tag', ['smoke_space', 'smoke_time'])
def test_smoke_super_model(trees, tag):
    gz, m = sub(tag), META[tag]
    shape = tuple(m['shape'])
    net = trees['Unet3D'](dim=8, dim_mults=(1, 2), channels=82, resnet_groups=4, init_kernel_size=3)
    kw = dict(loss_layer_weight=torch.from_numpy(gz['lw']), is_condition_control=m['is_condition_control'], is_condition_pad=True, is_wavelet=True,
              is_super_model=True, wave_type='bior1.3', pad_mode='zero', padded_shape=m['padded_shape'], ori_shape=None, image_size=shape[-1],
              frames=shape[1], timesteps=1000, sampling_timesteps=3, loss_type='l2', ddim_sampling_eta=1.)
    dif = trees['GD2'](net, **kw)
    dif.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in gz.items() if k.startswith('w::')}, strict=True)
    dif = dif.to(DEV)
    x0, t, noise = (torch.from_numpy(gz[k]).to(DEV) for k in ('x0', 't', 'noise'))
    loss = dif.p_losses(x0, t, noise=noise)
    assert abs(loss.item() - float(gz['loss'])) < TOL * abs(float(gz['loss']))
    loss.backward()
    check_grad_norms(dif, gz)
    low, init = torch.from_numpy(gz['ddim_low']).to(DEV), torch.from_numpy(gz['ddim_init']).to(DEV)
    control = torch.from_numpy(gz['ddim_control']).to(DEV) if 'ddim_control' in gz else None
    seq = iter([torch.from_numpy(gz[f'ddim_noise_{i}']).to(DEV) for i in range(int(gz['ddim_n_noise']))])
    dif.sample_noise = lambda shp, device: next(seq)
    out = dif.sample(batch_size=shape[0], N_upsample=m['n_up'], init=init, control=control, low=low)
    print(tag, 'ddim3 chain vs reference rel-L2', rel_l2(out, gz['ddim_out']))
    assert rel_l2(out, gz['ddim_out']) < DDIM3_TOL
    if 'ddpm3_out' in gz:
        dif3 = trees['GD2'](dif.model, **{**kw, 'timesteps': 3, 'sampling_timesteps': None}).to(DEV)
        seq3 = iter([torch.from_numpy(gz[f'ddpm3_noise_{i}']).to(DEV) for i in range(int(gz['ddpm3_n_noise']))])
        dif3.sample_noise = lambda shp, device: next(seq3)
        out = dif3.sample(batch_size=shape[0], N_upsample=m['n_up'], init=init, control=control, low=low)
        print(tag, 'ddpm3 chain vs reference rel-L2', rel_l2(out, gz['ddpm3_out']))
        assert rel_l2(out, gz['ddpm3_out']) < CHAIN_TOL
        assert torch.equal(out[:, :, 40:80].cpu(), torch.from_numpy(gz['ddim_low']))     # p_sample_loop re-imposes the low-resolution channels


def test_burgers_super_model_loss_and_grads(trees):
    gz, kw = sub('burgers'), dict(META['burgers'])
    kw['seq_length'] = tuple(kw['seq_length'])
    net = trees['Unet2D'](dim=8, dim_mults=(1, 2, 4), channels=17, resnet_block_groups=1)
    dif = trees['GD1'](net, loss_layer_weight=torch.from_numpy(gz['lw']), **kw)
    dif.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in gz.items() if k.startswith('w::')}, strict=True)
    dif = dif.to(DEV)
    x0, t, noise = (torch.from_numpy(gz[k]).to(DEV) for k in ('x0', 't', 'noise'))
    loss = dif.p_losses(x0, t, noise=noise)
    assert abs(loss.item() - float(gz['loss'])) < TOL * abs(float(gz['loss']))
    loss.backward()
    check_grad_norms(dif, gz)
