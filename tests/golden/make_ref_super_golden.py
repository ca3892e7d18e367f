"""Golden vectors for the super-resolution variants of the diffusion operators (rows D4-D7 with `low` conditioning; BASELINE
configs[4]) by importing the reference.  Build-container only:   python tests/golden/make_ref_super_golden.py
Writes tests/golden/ref_super_diffusion.npz (weights, inputs, injected noise, reference outputs -- data only)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_ref_golden as M  # noqa: E402

M.install_stubs()
torch.set_num_threads(8)
with M.cuda_default_args_on_cpu():
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion as GD1
from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
from ddpm.diffusion_2d import GaussianDiffusion as GD2

d, meta = {}, {}


def run_smoke(tag, is_condition_control, shape, padded_shape, n_up, seed):
    """space super-resolution (is_condition_control=True: +2 coefficients in h, w) or time super-resolution (+2 in t)"""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    net = Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2), channels=82, resnet_groups=4, init_kernel_size=3)
    M.randomise(net, g)
    lw = torch.rand(1, 1, 82, 1, 1, generator=g) * 5 + 1
    kw = dict(loss_layer_weight=lw, is_condition_control=is_condition_control, is_condition_pad=True, is_wavelet=True, is_super_model=True,
              wave_type='bior1.3', pad_mode='zero', padded_shape=padded_shape, ori_shape=None, image_size=shape[-1], frames=shape[1],
              timesteps=1000, sampling_timesteps=3, loss_type='l2', ddim_sampling_eta=1.)
    dif = GD2(net, **kw)
    x0 = torch.randn(*shape, generator=g) * 0.5
    t = torch.tensor([31, 777])[:shape[0]]
    noise = torch.randn(*shape, generator=g)
    loss = dif.p_losses(x0.clone(), t, noise=noise.clone())
    loss.backward()
    r = dict(x0=x0.numpy(), t=t.numpy(), noise=noise.numpy(), loss=loss.detach().numpy(), lw=lw.numpy())
    r.update(M.sd_np(dif)); r.update(M.grads_np(dif))
    low = torch.randn(shape[0], shape[1], 40, shape[3], shape[4], generator=g) * 0.3
    init = torch.randn(shape[0], shape[1], shape[3], shape[4], generator=g)
    control = torch.randn(shape[0], shape[1], 16, shape[3], shape[4], generator=g) if is_condition_control else None
    tape = M.NoiseTape(torch.Generator().manual_seed(seed + 100))
    with M.taped(tape):
        smp = dif.sample(batch_size=shape[0], N_upsample=n_up, init=init, control=control, low=low)
    r.update(ddim_low=low.numpy(), ddim_init=init.numpy(), ddim_out=smp.numpy(), ddim_n_noise=np.array(len(tape.tape)))
    if control is not None:
        r['ddim_control'] = control.numpy()
    for i, n in enumerate(tape.tape):
        r[f'ddim_noise_{i}'] = n.numpy()
    if not is_condition_control:
        for k, v in r.items():
            d[f'{tag}::{k}'] = v
        meta[tag] = dict(is_condition_control=is_condition_control, shape=list(shape), padded_shape=padded_shape, n_up=n_up)
        return
    dif3 = GD2(net, **{**kw, 'timesteps': 3, 'sampling_timesteps': None})
    tape = M.NoiseTape(torch.Generator().manual_seed(seed + 200))
    dif3.sample_noise = lambda shp, device, _tp=tape: _tp.randn(tuple(shp))
    with M.taped(tape):
        smp = dif3.sample(batch_size=shape[0], N_upsample=n_up, init=init, control=control, low=low)
    r.update(ddpm3_out=smp.numpy(), ddpm3_n_noise=np.array(len(tape.tape)))
    for i, n in enumerate(tape.tape):
        r[f'ddpm3_noise_{i}'] = n.numpy()
    for k, v in r.items():
        d[f'{tag}::{k}'] = v
    meta[tag] = dict(is_condition_control=is_condition_control, shape=list(shape), padded_shape=padded_shape, n_up=n_up)


# space SR: w = 10 -> level log2(40/10) = 2 ; time SR: f = 6 -> level log2(24/6) = 2 (small tensors keep the fixture small)
run_smoke('smoke_space', True, (1, 4, 82, 10, 10), [[3, 34, 34], [3, 16, 16], [3, 6, 6]], 2, 21)
run_smoke('smoke_time', False, (1, 6, 82, 8, 8), [[18, 6, 6], [9, 6, 6], [3, 6, 6]], 2, 22)

# ---- Burgers super-resolution model: 17 channels (8 fine + 8 low + condition), nx = 32 -> level log2(64/32) = 1
g = torch.Generator().manual_seed(23)
torch.manual_seed(23)
net = Unet2D(dim=8, dim_mults=(1, 2, 4), channels=17, resnet_block_groups=1)
M.randomise(net, g)
lw = torch.rand(1, 17, 1, 1, generator=g) + 0.5
kw = dict(seq_length=(32, 32), is_wavelet=True, pad_mode='periodization', wave_type='bior2.4', padded_shape=[[41, 60], [21, 28]],
          ori_shape=[[81, 120], [41, 56]], is_super_model=True, upsample_t=1, upsample_x=1, timesteps=1000, sampling_timesteps=3,
          ddim_sampling_eta=1., loss_layer_weight=lw, is_condition_pad=True, is_condition_u0=True, is_condition_uT=True, is_condition_f=True)
dif = GD1(net, **kw)
x0 = torch.randn(2, 17, 32, 32, generator=g) * 0.5
t = torch.tensor([9, 640])
noise = torch.randn(2, 17, 32, 32, generator=g)
loss = dif.p_losses(x0.clone(), t, noise=noise.clone())
loss.backward()
r = dict(x0=x0.numpy(), t=t.numpy(), noise=noise.numpy(), loss=loss.detach().numpy(), lw=lw.numpy())
r.update(M.sd_np(dif)); r.update(M.grads_np(dif))
for k, v in r.items():
    d[f'burgers::{k}'] = v
meta['burgers'] = {k: v for k, v in kw.items() if k != 'loss_layer_weight'}

np.savez_compressed(os.path.join(HERE, 'ref_super_diffusion.npz'), **d)
with open(os.path.join(HERE, 'ref_super_manifest.json'), 'w') as f:
    json.dump(meta, f, indent=1)
print('ref_super_diffusion.npz', os.path.getsize(os.path.join(HERE, 'ref_super_diffusion.npz')))
