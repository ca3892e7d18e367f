"""Generate golden vectors for the U-Net / diffusion rows of SURVEY.md section 8 by importing the reference.

Build-container only (needs /root/reference). Usage:
    python tests/golden/make_ref_golden.py
Writes tests/golden/ref_*.npz and ref_manifest.json. Only *data* (inputs, weights, expected outputs) is
written; no reference source travels.

The reference imports several packages that are not installed here. Minimal stand-ins are created in a
temporary directory (einops_exts helpers, the rotary embedding module restated from its published
definition, and empty shells for packages that are imported but never called on this path).
"""
import contextlib
import json
import os
import sys
import tempfile
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
_RANDN = torch.randn

STUBS = {
    'einops_exts/__init__.py': '''
        from einops import rearrange
        def rearrange_many(ts, pattern, **kw):
            return tuple(rearrange(t, pattern, **kw) for t in ts)
        def check_shape(t, *a, **k):
            return t
    ''',
    'rotary_embedding_torch/__init__.py': '''
        import torch
        from torch import nn
        class RotaryEmbedding(nn.Module):
            def __init__(self, dim, theta=10000):
                super().__init__()
                f = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
                self.freqs = nn.Parameter(f, requires_grad=False)
            def rotate_queries_or_keys(self, t, seq_dim=-2):
                n = t.shape[seq_dim]
                pos = torch.arange(n, device=t.device, dtype=self.freqs.dtype)
                ang = pos[:, None] * self.freqs[None, :]
                ang = ang.repeat_interleave(2, dim=-1)
                x = t.reshape(*t.shape[:-1], -1, 2)
                rot = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(t.shape)
                return t * ang.cos() + rot * ang.sin()
    ''',
    'torchvision/__init__.py': 'from . import transforms\nutils = None\n',
    'torchvision/transforms.py': 'def ToTensor():\n    return None\n',
    'ema_pytorch/__init__.py': 'EMA = None\n',
    'tensorboardX/__init__.py': 'SummaryWriter = None\n',
    'IPython/__init__.py': 'embed = None\n',
    'pytorch_wavelets/__init__.py': 'DWTForward = DWTInverse = DWT1DForward = DWT1DInverse = None\n',
    'pywt/__init__.py': 'wavedec = waverec = None\n',
    'ptwt/__init__.py': '',
    'h5py/__init__.py': '',
}


def install_stubs():
    d = tempfile.mkdtemp(prefix='wdno_stubs_')
    for rel, src in STUBS.items():
        p = os.path.join(d, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, 'w') as f:
            f.write(textwrap.dedent(src))
    sys.path[:0] = [d, os.path.join(REF, 'burgers'), os.path.join(REF, 'smoke')]


@contextlib.contextmanager
def cuda_default_args_on_cpu():
    """diffusion_1d.py:51 builds a default arg with device='cuda' at import time."""
    orig = torch.tensor

    def patched(*a, **k):
        if str(k.get('device', '')).startswith('cuda'):
            k['device'] = 'cpu'
        return orig(*a, **k)
    torch.tensor = patched
    try:
        yield
    finally:
        torch.tensor = orig


def randomise(model, gen, scale=0.05):
    """Perturb every parameter (norm gains/biases default to 1/0 and would hide missing-affine bugs)."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('freqs'):
                continue
            p.add_(scale * torch.randn(p.shape, generator=gen))


def sd_np(model, prefix='w::'):
    return {prefix + k: v.detach().cpu().clone().numpy() for k, v in model.state_dict().items()}


def grads_np(model, prefix='g::', full_below=6000):
    """Full gradient arrays for small parameters, L2 norms ('gn::key') for every parameter (keeps fixtures small)."""
    out = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        gr = p.grad.detach().cpu()
        out['gn::' + k] = gr.double().norm().numpy()
        if gr.numel() <= full_below:
            out[prefix + k] = gr.numpy()
    return out


class NoiseTape:
    """Replaces torch.randn / randn_like in a reference module so the drawn noise can be recorded."""
    def __init__(self, gen):
        self.gen = gen
        self.tape = []

    def randn(self, *shape, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        x = _RANDN(shape, generator=self.gen)
        self.tape.append(x.clone())
        return x

    def randn_like(self, t, **kw):
        return self.randn(tuple(t.shape))


@contextlib.contextmanager
def taped(tape):
    o1, o2 = torch.randn, torch.randn_like
    torch.randn, torch.randn_like = tape.randn, tape.randn_like
    try:
        yield
    finally:
        torch.randn, torch.randn_like = o1, o2


def main():
    install_stubs()
    torch.set_num_threads(8)
    with cuda_default_args_on_cpu():
        from ddpm_burgers.unet import Unet2D
        from ddpm_burgers.diffusion_1d import GaussianDiffusion as GD1
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion as GD2
    manifest = {}

    # ------------------------------------------------------------------ (1) smoke U-Net, tiny
    g = torch.Generator().manual_seed(1)
    torch.manual_seed(1)
    net = Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2, 4), channels=6, resnet_groups=4)
    randomise(net, g)
    x = torch.randn(2, 4, 6, 8, 8, generator=g, requires_grad=True)
    t = torch.tensor([3, 977])
    gout = torch.randn(2, 4, 6, 8, 8, generator=g)
    out = net(x, t)
    (out * gout).sum().backward()
    d = dict(x=x.detach().numpy(), t=t.numpy(), gout=gout.numpy(), out=out.detach().numpy(), gx=x.grad.numpy())
    d.update(sd_np(net)); d.update(grads_np(net))
    np.savez_compressed(os.path.join(HERE, 'ref_unet3d_tiny.npz'), **d)
    manifest['unet3d_tiny'] = dict(dim=8, dim_mults=[1, 2, 4], channels=6, resnet_groups=4)

    # ------------------------------------------------------------------ (2) Burgers U-Net, tiny
    g = torch.Generator().manual_seed(2)
    torch.manual_seed(2)
    net = Unet2D(dim=8, dim_mults=(1, 2, 4), channels=9, resnet_block_groups=1)
    randomise(net, g)
    x = torch.randn(2, 9, 16, 16, generator=g, requires_grad=True)
    t = torch.tensor([0, 640])
    gout = torch.randn(2, 9, 16, 16, generator=g)
    out = net(x, t)
    (out * gout).sum().backward()
    d = dict(x=x.detach().numpy(), t=t.numpy(), gout=gout.numpy(), out=out.detach().numpy(), gx=x.grad.numpy())
    d.update(sd_np(net)); d.update(grads_np(net))
    np.savez_compressed(os.path.join(HERE, 'ref_unet2d_tiny.npz'), **d)
    manifest['unet2d_tiny'] = dict(dim=8, dim_mults=[1, 2, 4], channels=9, resnet_block_groups=1)

    # a second Burgers variant with grouped norm (groups=4) on a non-square plane
    g = torch.Generator().manual_seed(22)
    torch.manual_seed(22)
    net = Unet2D(dim=8, dim_mults=(1, 2), channels=3, resnet_block_groups=4)
    randomise(net, g)
    x = torch.randn(1, 3, 8, 12, generator=g, requires_grad=True)
    t = torch.tensor([123])
    gout = torch.randn(1, 3, 8, 12, generator=g)
    out = net(x, t)
    (out * gout).sum().backward()
    d = dict(x=x.detach().numpy(), t=t.numpy(), gout=gout.numpy(), out=out.detach().numpy(), gx=x.grad.numpy())
    d.update(sd_np(net)); d.update(grads_np(net))
    np.savez_compressed(os.path.join(HERE, 'ref_unet2d_g4.npz'), **d)
    manifest['unet2d_g4'] = dict(dim=8, dim_mults=[1, 2], channels=3, resnet_block_groups=4)

    # ------------------------------------------------------------------ (3) smoke diffusion operator
    g = torch.Generator().manual_seed(3)
    torch.manual_seed(3)
    net = Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2, 4), channels=42, resnet_groups=4)
    randomise(net, g)
    lw = torch.rand(1, 1, 42, 1, 1, generator=g) * 10 + 1
    kw = dict(loss_layer_weight=lw, is_condition_control=True, is_condition_pad=True, is_wavelet=True,
              is_super_model=False, wave_type='bior1.3', pad_mode='zero', padded_shape=(3, 6, 6),
              ori_shape=(4, 8, 8), image_size=8, frames=4, timesteps=1000, sampling_timesteps=4,
              loss_type='l2', ddim_sampling_eta=1.)
    dif = GD2(net, **kw)
    x0 = torch.randn(2, 4, 42, 8, 8, generator=g) * 0.5
    t = torch.tensor([17, 803])
    noise = torch.randn(2, 4, 42, 8, 8, generator=g)
    loss = dif.p_losses(x0.clone(), t, noise=noise.clone())
    loss.backward()
    d = dict(x0=x0.numpy(), t=t.numpy(), noise=noise.numpy(), loss=loss.detach().numpy(), lw=lw.numpy())
    d.update(sd_np(dif)); d.update(grads_np(dif))
    # q_sample alone
    d['q_sample'] = dif.q_sample(x0, t, noise).numpy()
    # p_sample at a few t with injected noise
    xt = torch.randn(2, 4, 42, 8, 8, generator=g)
    for tt in (0, 500, 999):
        tape = NoiseTape(torch.Generator().manual_seed(100 + tt))
        dif.sample_noise = lambda shape, device, _tp=tape: _tp.randn(tuple(shape))
        pred, xs = dif.p_sample(xt.shape, xt.clone(), tt)
        d[f'psample_{tt}_pred'] = pred.numpy(); d[f'psample_{tt}_xstart'] = xs.numpy()
        d[f'psample_{tt}_noise'] = tape.tape[0].numpy() if tape.tape else np.zeros(0, np.float32)
    d['psample_xt'] = xt.numpy()
    # DDIM sampling, 4 steps, eta=1, with recorded noise
    init = torch.randn(2, 4, 8, 8, generator=g)
    control = torch.randn(2, 4, 16, 8, 8, generator=g)
    tape = NoiseTape(torch.Generator().manual_seed(7))
    with taped(tape):
        smp = dif.sample(batch_size=2, init=init, control=control)
    d['ddim_init'] = init.numpy(); d['ddim_control'] = control.numpy(); d['ddim_out'] = smp.numpy()
    for i, n in enumerate(tape.tape):
        d[f'ddim_noise_{i}'] = n.numpy()
    d['ddim_n_noise'] = np.array(len(tape.tape))
    # DDPM ancestral loop with T=5
    dif5 = GD2(net, **{**kw, 'timesteps': 5, 'sampling_timesteps': None})
    tape = NoiseTape(torch.Generator().manual_seed(8))
    dif5.sample_noise = lambda shape, device, _tp=tape: _tp.randn(tuple(shape))
    smp = dif5.sample(batch_size=2, init=init, control=control)
    d['ddpm5_out'] = smp.numpy()
    for i, n in enumerate(tape.tape):
        d[f'ddpm5_noise_{i}'] = n.numpy()
    d['ddpm5_n_noise'] = np.array(len(tape.tape))
    np.savez_compressed(os.path.join(HERE, 'ref_smoke_diffusion.npz'), **d)
    manifest['smoke_diffusion'] = dict(unet=dict(dim=8, dim_mults=[1, 2, 4], channels=42, resnet_groups=4),
                                       diffusion={k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items() if k != 'loss_layer_weight'})

    # ------------------------------------------------------------------ (4) Burgers diffusion operator
    g = torch.Generator().manual_seed(4)
    torch.manual_seed(4)
    net = Unet2D(dim=8, dim_mults=(1, 2, 4), channels=9, resnet_block_groups=1)
    randomise(net, g)
    lw = torch.rand(1, 9, 1, 1, generator=g) + 0.5
    kw = dict(seq_length=(16, 16), is_wavelet=True, pad_mode='periodization', wave_type='bior2.4',
              padded_shape=[11, 14], ori_shape=[20, 28], timesteps=1000, sampling_timesteps=4,
              ddim_sampling_eta=1., loss_layer_weight=lw, is_condition_pad=True, is_condition_u0=True,
              is_condition_uT=False, is_condition_f=True)
    dif = GD1(net, **kw)
    x0 = torch.randn(2, 9, 16, 16, generator=g) * 0.5
    t = torch.tensor([5, 911])
    noise = torch.randn(2, 9, 16, 16, generator=g)
    loss = dif.p_losses(x0.clone(), t, noise=noise.clone())
    loss.backward()
    d = dict(x0=x0.numpy(), t=t.numpy(), noise=noise.numpy(), loss=loss.detach().numpy(), lw=lw.numpy())
    d.update(sd_np(dif)); d.update(grads_np(dif))
    d['alphas'] = dif.alphas.numpy(); d['alphas_prev'] = dif.alphas_prev.numpy()
    u_init = torch.randn(2, 8, 16, generator=g)
    f = torch.randn(2, 4, 16, 16, generator=g)
    tape = NoiseTape(torch.Generator().manual_seed(9))
    with taped(tape):
        smp = dif.sample(batch_size=2, u_init=u_init, f=f)
    d['ddim_u_init'] = u_init.numpy(); d['ddim_f'] = f.numpy(); d['ddim_out'] = smp.numpy()
    for i, n in enumerate(tape.tape):
        d[f'ddim_noise_{i}'] = n.numpy()
    d['ddim_n_noise'] = np.array(len(tape.tape))
    dif5 = GD1(net, **{**kw, 'timesteps': 5, 'sampling_timesteps': None})
    tape = NoiseTape(torch.Generator().manual_seed(10))
    with taped(tape):
        smp = dif5.sample(batch_size=2, u_init=u_init, f=f)
    d['ddpm5_out'] = smp.numpy()
    for i, n in enumerate(tape.tape):
        d[f'ddpm5_noise_{i}'] = n.numpy()
    d['ddpm5_n_noise'] = np.array(len(tape.tape))
    # all condition flags at once (uT too) : loss only
    difT = GD1(net, **{**kw, 'is_condition_uT': True})
    d['loss_all_cond'] = difT.p_losses(x0.clone(), t, noise=noise.clone()).detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'ref_burgers_diffusion.npz'), **d)
    manifest['burgers_diffusion'] = dict(unet=dict(dim=8, dim_mults=[1, 2, 4], channels=9, resnet_block_groups=1),
                                         diffusion={k: v for k, v in kw.items() if k != 'loss_layer_weight'})

    # ------------------------------------------------------------------ (5) schedule buffers
    d = {}
    dummy2 = Unet2D(dim=8, dim_mults=(1,), channels=2)
    for sched in ('cosine', 'linear'):
        m = GD1(dummy2, seq_length=(8, 8), beta_schedule=sched, ori_shape=[8, 8])
        for k, v in m.state_dict().items():
            if not k.startswith('model.'):
                d[f'burgers_{sched}::{k}'] = v.numpy()
    dummy3 = Unet3D_with_Conv3D(dim=8, dim_mults=(1,), channels=2)
    for sched in ('sigmoid', 'cosine', 'linear'):
        m = GD2(dummy3, None, False, False, True, False, 'bior1.3', 'zero', None, None, image_size=8, frames=2, beta_schedule=sched)
        for k, v in m.state_dict().items():
            if not k.startswith('model.'):
                d[f'smoke_{sched}::{k}'] = v.numpy()
    np.savez_compressed(os.path.join(HERE, 'ref_schedules.npz'), **d)

    # ------------------------------------------------------------------ (6) trainer math: 3 optimiser steps (T1/T2 rows)
    # burgers/ddpm_burgers/train_diffusion.py:117-118,203-216 ; smoke/ddpm/diffusion_2d.py:1159-1161,1277-1293
    d = {}
    g = torch.Generator().manual_seed(6)
    torch.manual_seed(6)
    net = Unet2D(dim=8, dim_mults=(1, 2), channels=9, resnet_block_groups=1)
    randomise(net, g)
    dif = GD1(net, seq_length=(8, 8), padded_shape=[6, 7], ori_shape=[10, 14], loss_layer_weight=torch.ones(1, 9, 1, 1),
              is_condition_pad=True, is_condition_u0=True, is_condition_f=True)
    d.update(sd_np(dif, 'w0::'))
    opt = torch.optim.Adam(dif.parameters(), lr=1e-4, betas=(0.9, 0.99))
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10000)
    for step in range(3):
        x0 = torch.randn(4, 9, 8, 8, generator=g) * 0.5
        t = torch.randint(0, 1000, (4,), generator=g)
        noise = torch.randn(4, 9, 8, 8, generator=g)
        loss = dif.p_losses(x0.clone(), t, noise=noise.clone())
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(dif.parameters(), 1.0)
        opt.step(); opt.zero_grad(); sch.step()
        d[f's{step}_x0'] = x0.numpy(); d[f's{step}_t'] = t.numpy(); d[f's{step}_noise'] = noise.numpy()
        d[f's{step}_loss'] = loss.detach().numpy(); d[f's{step}_gnorm'] = gn.numpy()
    d.update(sd_np(dif, 'w3::'))
    np.savez_compressed(os.path.join(HERE, 'ref_train_burgers.npz'), **d)
    manifest['train_burgers'] = dict(unet=dict(dim=8, dim_mults=[1, 2], channels=9, resnet_block_groups=1),
                                     padded_shape=[6, 7], seq_length=[8, 8])

    # ------------------------------------------------------------------ (7) state_dict manifests of the full-size models
    with torch.device('meta'):
        big2 = Unet2D(dim=128, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
        big3 = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    manifest['state_dict_unet2d_full'] = {k: list(v.shape) for k, v in big2.state_dict().items()}
    manifest['state_dict_unet3d_full'] = {k: list(v.shape) for k, v in big3.state_dict().items()}
    manifest['n_params_unet2d_full'] = sum(p.numel() for p in big2.parameters())
    manifest['n_params_unet3d_full'] = sum(p.numel() for p in big3.parameters() if p.requires_grad)
    manifest['diffusion_buffers'] = [k for k in dif.state_dict().keys() if not k.startswith('model.')]

    with open(os.path.join(HERE, 'ref_manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=1, default=str)
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == '__main__':
    main()
