"""Pins oracle/diffusion_ref.py against outputs of the imported reference."""
import functools

import numpy as np
import pytest
import torch

from oracle import diffusion_ref as D
from oracle import unet_ref as U
from tests.helpers import load_npz, manifest, noise_seq, rel_l2, weights

M = manifest()


@pytest.mark.parametrize('tree,kind', [('burgers', 'cosine'), ('burgers', 'linear'), ('smoke', 'sigmoid'), ('smoke', 'cosine'), ('smoke', 'linear')])
def test_schedule_buffers(tree, kind):
    g = load_npz('ref_schedules.npz')
    buf = D.make_buffers(kind, 1000)
    keys = [k.split('::')[1] for k in g.files if k.startswith(f'{tree}_{kind}::')]
    assert sorted(keys) == sorted(buf.keys()) == sorted(M['diffusion_buffers'])
    for k in keys:
        ref = torch.from_numpy(g[f'{tree}_{kind}::{k}'])
        assert buf[k].dtype == torch.float32 and buf[k].shape == (1000,)
        assert torch.allclose(buf[k], ref, rtol=1e-6, atol=1e-7), k


def _smoke_setup():
    g = load_npz('ref_smoke_diffusion.npz')
    c = M['smoke_diffusion']
    sd = weights(g, 'w::model.', requires_grad=True)
    u = c['unet']
    model = lambda x, t: U.unet3d_forward(sd, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_groups'])
    return g, c['diffusion'], sd, model


def test_smoke_p_losses_and_grads():
    g, c, sd, model = _smoke_setup()
    buf = D.make_buffers('sigmoid', 1000)
    x0, t, noise = (torch.from_numpy(g[k]) for k in ('x0', 't', 'noise'))
    assert torch.allclose(D.q_sample(buf, x0, t, noise), torch.from_numpy(g['q_sample']), rtol=1e-6, atol=1e-6)
    loss = D.smoke_p_losses(model, buf, x0, t, noise, padded_shape=c['padded_shape'], loss_layer_weight=torch.from_numpy(g['lw']))
    assert abs(loss.item() - float(g['loss'])) < 2e-6 * abs(float(g['loss']))
    loss.backward()
    for k in g.files:
        if k.startswith('gn::model.'):
            ref = float(g[k]); got = sd[k[len('gn::model.'):]].grad.double().norm().item()
            assert abs(got - ref) <= 5e-5 * max(ref, 1e-7), (k, got, ref)


def test_smoke_sampling():
    g, c, sd, model = _smoke_setup()
    buf = D.make_buffers('sigmoid', 1000)
    with torch.no_grad():
        xt = torch.from_numpy(g['psample_xt'])
        for tt in (0, 500, 999):
            nz = torch.from_numpy(g[f'psample_{tt}_noise']) if tt > 0 else None
            pred, xs = D.smoke_p_sample(model, buf, xt.clone(), tt, nz)
            assert rel_l2(pred, g[f'psample_{tt}_pred']) < 5e-6 and rel_l2(xs, g[f'psample_{tt}_xstart']) < 5e-6
        kw = dict(padded_shape=c['padded_shape'], init=torch.from_numpy(g['ddim_init']), control=torch.from_numpy(g['ddim_control']))
        out = D.smoke_ddim_sample(model, buf, noise_seq(g, 'ddim'), 1000, 4, 1.0, **kw)
        assert rel_l2(out, g['ddim_out']) < 1e-5
        out = D.smoke_p_sample_loop(model, D.make_buffers('sigmoid', 5), noise_seq(g, 'ddpm5'), 5, **kw)
        assert rel_l2(out, g['ddpm5_out']) < 1e-5


def _burgers_setup():
    g = load_npz('ref_burgers_diffusion.npz')
    c = M['burgers_diffusion']
    sd = weights(g, 'w::model.', requires_grad=True)
    u = c['unet']
    model = lambda x, t: U.unet2d_forward(sd, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_block_groups'])
    return g, c['diffusion'], sd, model


def test_burgers_p_losses_and_grads():
    g, c, sd, model = _burgers_setup()
    buf = D.make_buffers('cosine', 1000)
    x0, t, noise = (torch.from_numpy(g[k]) for k in ('x0', 't', 'noise'))
    flags = dict(pad=True, u0=True, uT=False, f=True)
    lw = torch.from_numpy(g['lw'])
    loss = D.burgers_p_losses(model, buf, x0, t, noise, padded_shape=c['padded_shape'], loss_layer_weight=lw, flags=flags)
    assert abs(loss.item() - float(g['loss'])) < 2e-6 * abs(float(g['loss']))
    loss.backward()
    for k in g.files:
        if k.startswith('gn::model.'):
            ref = float(g[k]); got = sd[k[len('gn::model.'):]].grad.double().norm().item()
            assert abs(got - ref) <= 5e-5 * max(ref, 1e-7), (k, got, ref)
    with torch.no_grad():
        l2 = D.burgers_p_losses(model, buf, x0, t, noise, padded_shape=c['padded_shape'], loss_layer_weight=lw,
                                flags=dict(flags, uT=True))
    assert abs(l2.item() - float(g['loss_all_cond'])) < 2e-6 * abs(float(g['loss_all_cond']))


def test_burgers_sampling():
    g, c, sd, model = _burgers_setup()
    flags = dict(pad=True, u0=True, uT=False, f=True)
    with torch.no_grad():
        kw = dict(padded_shape=c['padded_shape'], flags=flags, u0=torch.from_numpy(g['ddim_u_init']), f=torch.from_numpy(g['ddim_f']))
        out = D.burgers_ddim_sample(model, D.make_buffers('cosine', 1000), noise_seq(g, 'ddim'), 1000, 4, 1.0, **kw)
        assert rel_l2(out, g['ddim_out']) < 1e-5
        out = D.burgers_p_sample_loop(model, D.make_buffers('cosine', 5), noise_seq(g, 'ddpm5'), 5, **kw)
        assert rel_l2(out, g['ddpm5_out']) < 1e-5


def test_three_optimizer_steps_oracle_vs_reference():
    """T1 row: the oracle p_losses + torch Adam/clip/cosine schedule reproduces the reference's 3-step trajectory."""
    g = load_npz('ref_train_burgers.npz')
    c = M['train_burgers']
    sd = weights(g, 'w0::model.', requires_grad=True)
    params = [v for v in sd.values() if v.requires_grad]
    u = c['unet']
    model = lambda x, t: U.unet2d_forward(sd, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_block_groups'])
    buf = D.make_buffers('cosine', 1000)
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.99))
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10000)
    for step in range(3):
        x0, t, noise = (torch.from_numpy(g[f's{step}_{k}']) for k in ('x0', 't', 'noise'))
        loss = D.burgers_p_losses(model, buf, x0, t, noise, padded_shape=c['padded_shape'], loss_layer_weight=torch.ones(1, 9, 1, 1),
                                  flags=dict(pad=True, u0=True, f=True))
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step(); sch.step()
        assert abs(loss.item() - float(g[f's{step}_loss'])) < 1e-5 * abs(float(g[f's{step}_loss']))
        assert abs(gn.item() - float(g[f's{step}_gnorm'])) < 1e-4 * float(g[f's{step}_gnorm'])
    assert not np.array_equal(g['w0::model.init_conv.weight'], g['w3::model.init_conv.weight'])
