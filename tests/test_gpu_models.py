"""End-to-end parity of the HIP module trees against the reference golden vectors and the CPU oracle (GPU box only).

The modules are imported through the reference's own import paths (ddpm_burgers.unet, ddpm.diffusion_2d, ...) from the
drop-in trees wdno_amd/burgers and wdno_amd/smoke. Tolerance: rel-L2 < 1e-5 (north-star bar for fp32 fields)."""
import sys

import pytest
import torch

from tests.helpers import load_npz, manifest, noise_seq, rel_l2, weights

pytestmark = pytest.mark.gpu
DEV = 'cuda'
M = manifest()
TOL = 1e-5
# Sampled chains against the reference's fp32 outputs: 1e-5 where the chain is well conditioned (ancestral chains, Burgers DDIM).
# Round 3: the 4-step smoke DDIM chain from t = 999 is held to the same 1e-5 (measured 3.9e-6; 8 more seeds in
# tests/test_gpu_fullsize.py::test_ddim_chains_from_t999_over_seeds, worst 9.0e-6). It used to sit at 5.3e-5 for ONE reason, found by
# tracing the chain step by step against the reference on both hosts: c = sqrt(1 - a' - sigma^2) of the first step cancels to 1.4e-6
# from terms ~1, torch's CPU sqrt is not correctly rounded and rounds sigma differently on the GPU box's host than in the container the
# fixtures were made in, so the SAME reference code gives c = 1.245e-3 there and 1.196e-3 here. The coefficients are now evaluated
# with correctly rounded fp32 arithmetic (wdno_amd.diffusion_core.ddim_coefficients, tools/ddim_coef_probe.py).
CHAIN_TOL = 1e-5
SMOKE_DDIM4_TOL = 1e-5


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


def load_into(module, npz, prefix):
    module.load_state_dict(weights(npz, prefix), strict=True)
    return module.to(DEV)


def check_unet(net, gz, tol=TOL):
    x = torch.from_numpy(gz['x']).to(DEV).requires_grad_(True)
    out = net(x, torch.from_numpy(gz['t']).to(DEV))
    assert rel_l2(out.detach(), gz['out']) < tol, 'forward'
    (out * torch.from_numpy(gz['gout']).to(DEV)).sum().backward()
    assert rel_l2(x.grad, gz['gx']) < 3 * tol, 'input grad'
    params = dict(net.named_parameters())
    worst = 0.0
    for k in gz.files:
        if k.startswith('gn::'):
            ref = float(gz[k]); got = params[k[4:]].grad.double().norm().item()
            assert abs(got - ref) <= 1e-4 * max(ref, 1e-6), (k, got, ref)
        if k.startswith('g::'):
            e = rel_l2(params[k[3:]].grad, gz[k])
            worst = max(worst, e)
            assert e < 1e-4, (k, e)
    return worst


def test_unet3d_tiny_vs_reference(trees):
    c = M['unet3d_tiny']
    net = trees['Unet3D'](dim=c['dim'], dim_mults=tuple(c['dim_mults']), channels=c['channels'], resnet_groups=c['resnet_groups'])
    gz = load_npz('ref_unet3d_tiny.npz')
    check_unet(load_into(net, gz, 'w::'), gz)


def test_unet2d_tiny_vs_reference(trees):
    c = M['unet2d_tiny']
    net = trees['Unet2D'](dim=c['dim'], dim_mults=tuple(c['dim_mults']), channels=c['channels'], resnet_block_groups=c['resnet_block_groups'])
    gz = load_npz('ref_unet2d_tiny.npz')
    check_unet(load_into(net, gz, 'w::'), gz)


def test_unet2d_grouped_vs_reference(trees):
    c = M['unet2d_g4']
    net = trees['Unet2D'](dim=c['dim'], dim_mults=tuple(c['dim_mults']), channels=c['channels'], resnet_block_groups=c['resnet_block_groups'])
    gz = load_npz('ref_unet2d_g4.npz')
    check_unet(load_into(net, gz, 'w::'), gz)


def _smoke(trees):
    gz = load_npz('ref_smoke_diffusion.npz')
    c = M['smoke_diffusion']
    u, d = c['unet'], dict(c['diffusion'])
    net = trees['Unet3D'](dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_groups=u['resnet_groups'])
    d['padded_shape'] = tuple(d['padded_shape']); d['ori_shape'] = tuple(d['ori_shape'])
    return gz, d, net


def test_smoke_diffusion_loss_and_grads(trees):
    gz, d, net = _smoke(trees)
    dif = trees['GD2'](net, loss_layer_weight=torch.from_numpy(gz['lw']), **d)
    load_into(dif, gz, 'w::')
    x0, t, noise = (torch.from_numpy(gz[k]).to(DEV) for k in ('x0', 't', 'noise'))
    assert (dif.q_sample(x0, t, noise).cpu() - torch.from_numpy(gz['q_sample'])).abs().max() < 1e-6
    loss = dif.p_losses(x0, t, noise=noise)
    assert abs(loss.item() - float(gz['loss'])) < TOL * abs(float(gz['loss']))
    loss.backward()
    params = dict(dif.named_parameters())
    for k in gz.files:
        if k.startswith('gn::'):
            ref = float(gz[k]); got = params[k[4:]].grad.double().norm().item()
            assert abs(got - ref) <= 2e-4 * max(ref, 1e-7), (k, got, ref)


def test_smoke_sampling(trees):
    gz, d, net = _smoke(trees)
    dif = load_into(trees['GD2'](net, loss_layer_weight=torch.from_numpy(gz['lw']), **d), gz, 'w::')
    xt = torch.from_numpy(gz['psample_xt']).to(DEV)
    for tt in (0, 500, 999):
        nz = torch.from_numpy(gz[f'psample_{tt}_noise']).to(DEV)
        dif.sample_noise = lambda shape, device, _n=nz: _n
        pred, xs = dif.p_sample(tuple(xt.shape), xt.clone(), tt)
        # x_start = c1 x - c2 eps_hat multiplies the U-Net's fp32 round-off by c2 = sqrt(1/ac_t - 1) (~1e2 near t = T)
        amp = max(1.0, float(dif.sqrt_recipm1_alphas_cumprod[tt]))
        assert rel_l2(pred, gz[f'psample_{tt}_pred']) < TOL and rel_l2(xs, gz[f'psample_{tt}_xstart']) < TOL * amp, (tt, amp)
    init, control = torch.from_numpy(gz['ddim_init']).to(DEV), torch.from_numpy(gz['ddim_control']).to(DEV)
    seq = iter([n.to(DEV) for n in noise_seq(gz, 'ddim')])
    dif.sample_noise = lambda shape, device: next(seq)
    out = dif.sample(batch_size=2, init=init, control=control)
    print('smoke ddim4 chain vs reference rel-L2', rel_l2(out, gz['ddim_out']))
    assert rel_l2(out, gz['ddim_out']) < SMOKE_DDIM4_TOL
    dif5 = trees['GD2'](dif.model, loss_layer_weight=torch.from_numpy(gz['lw']), **{**d, 'timesteps': 5, 'sampling_timesteps': None}).to(DEV)
    seq5 = iter([n.to(DEV) for n in noise_seq(gz, 'ddpm5')])
    dif5.sample_noise = lambda shape, device: next(seq5)
    out = dif5.sample(batch_size=2, init=init, control=control)
    print('smoke ddpm5 chain vs reference rel-L2', rel_l2(out, gz['ddpm5_out']))
    assert rel_l2(out, gz['ddpm5_out']) < CHAIN_TOL


def _burgers(trees, **over):
    gz = load_npz('ref_burgers_diffusion.npz')
    c = M['burgers_diffusion']
    u, d = c['unet'], dict(c['diffusion'])
    d.update(over)
    net = trees['Unet2D'](dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_block_groups=u['resnet_block_groups'])
    d['seq_length'] = tuple(d['seq_length'])
    dif = trees['GD1'](net, loss_layer_weight=torch.from_numpy(gz['lw']), **d)
    sd = {k: v for k, v in weights(gz, 'w::').items()}
    if over.get('timesteps'):
        sd = {k: v for k, v in sd.items() if k.startswith('model.')}
        dif.load_state_dict(sd, strict=False)
    else:
        dif.load_state_dict(sd, strict=True)
    return gz, dif.to(DEV)


def test_burgers_diffusion_loss_and_grads(trees):
    gz, dif = _burgers(trees)
    assert torch.allclose(dif.alphas, torch.from_numpy(gz['alphas'])) and torch.allclose(dif.alphas_prev, torch.from_numpy(gz['alphas_prev']))
    x0, t, noise = (torch.from_numpy(gz[k]).to(DEV) for k in ('x0', 't', 'noise'))
    loss = dif.p_losses(x0, t, noise=noise)
    assert abs(loss.item() - float(gz['loss'])) < TOL * abs(float(gz['loss']))
    loss.backward()
    params = dict(dif.named_parameters())
    for k in gz.files:
        if k.startswith('gn::'):
            ref = float(gz[k]); got = params[k[4:]].grad.double().norm().item()
            assert abs(got - ref) <= 2e-4 * max(ref, 1e-7), (k, got, ref)
    gz2, dif2 = _burgers(trees, is_condition_uT=True)
    l2 = dif2.p_losses(x0, t, noise=noise)
    assert abs(l2.item() - float(gz['loss_all_cond'])) < TOL * abs(float(gz['loss_all_cond']))


def test_burgers_sampling(trees):
    gz, dif = _burgers(trees)
    u_init, f = torch.from_numpy(gz['ddim_u_init']).to(DEV), torch.from_numpy(gz['ddim_f']).to(DEV)
    seq = iter([n.to(DEV) for n in noise_seq(gz, 'ddim')])
    dif.sample_noise = lambda shape, device: next(seq)
    out = dif.sample(batch_size=2, u_init=u_init, f=f)
    print('burgers ddim4 chain vs reference rel-L2', rel_l2(out, gz['ddim_out']))
    assert rel_l2(out, gz['ddim_out']) < CHAIN_TOL
    gz5, dif5 = _burgers(trees, timesteps=5, sampling_timesteps=None)
    seq5 = iter([n.to(DEV) for n in noise_seq(gz, 'ddpm5')])
    dif5.sample_noise = lambda shape, device: next(seq5)
    out = dif5.sample(batch_size=2, u_init=u_init, f=f)
    print('burgers ddpm5 chain vs reference rel-L2', rel_l2(out, gz['ddpm5_out']))
    assert rel_l2(out, gz['ddpm5_out']) < CHAIN_TOL


def test_three_optimizer_steps_vs_reference(trees):
    """T1 row: loss -> backward -> clip(1.0) -> Adam(1e-4, (0.9, 0.99)) -> CosineAnnealingLR(10000), three steps."""
    from wdno_amd.trainer import TrainStep, cosine_annealing_lr
    gz = load_npz('ref_train_burgers.npz')
    c = M['train_burgers']
    u = c['unet']
    net = trees['Unet2D'](dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_block_groups=u['resnet_block_groups'])
    dif = trees['GD1'](net, seq_length=tuple(c['seq_length']), padded_shape=c['padded_shape'], ori_shape=[10, 14],
                       loss_layer_weight=torch.ones(1, 9, 1, 1), is_condition_pad=True, is_condition_u0=True, is_condition_f=True)
    dif.load_state_dict(weights(gz, 'w0::'), strict=True)
    dif = dif.to(DEV)
    ts = TrainStep(dif, lr=1e-4, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=lambda b, s: cosine_annealing_lr(b, s, 10000), use_ema=False)
    from tests import arbiter as A
    le, ge, exact = A.burgers_train3(torch.float64)        # the same three steps by the oracle in fp64: the arbiter
    rel = lambda a, b: abs(a - b) / abs(b)
    for step in range(3):
        x0, t, noise = (torch.from_numpy(gz[f's{step}_{k}']).to(DEV) for k in ('x0', 't', 'noise'))
        loss, gn = ts.step_with(x0, t, noise)
        assert A.gate_or_bar(rel(loss.item(), le[step]), rel(float(gz[f's{step}_loss']), le[step])), (step, loss.item(), le[step])
        assert A.gate_or_bar(rel(gn.item(), ge[step]), rel(float(gz[f's{step}_gnorm']), ge[step])), (step, gn.item(), ge[step])
    w0 = {k[len('model.'):]: v for k, v in weights(gz, 'w0::').items() if k.startswith('model.')}
    ref = {k[len('model.'):]: v for k, v in weights(gz, 'w3::').items() if k.startswith('model.')}
    ours = {k[len('model.'):]: v.cpu() for k, v in dif.state_dict().items() if k.startswith('model.')}
    A.check_updates('T1 burgers', w0, ours, ref, exact)


def test_unet3d_real_width_vs_oracle(trees):
    """dim = 64 (the real smoke width) at a reduced grid: HIP vs the CPU oracle, forward and parameter gradients."""
    from oracle import unet_ref as U
    torch.manual_seed(0)
    net = trees['Unet3D'](dim=64, dim_mults=(1, 2, 4), channels=42)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(1, 6, 42, 16, 16)
    t = torch.tensor([321])
    go = torch.randn(1, 6, 42, 16, 16)
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith('freqs') else v) for k, v in sd.items()}
    out_ref = U.unet3d_forward(sdr, x, t, dim=64, dim_mults=(1, 2, 4), groups=8)
    (out_ref * go).sum().backward()
    net = net.to(DEV)
    out = net(x.to(DEV), t.to(DEV))
    assert rel_l2(out.detach(), out_ref.detach()) < TOL
    (out * go.to(DEV)).sum().backward()
    for k, p in net.named_parameters():
        if p.grad is not None:
            assert rel_l2(p.grad, sdr[k].grad) < 2e-4, k


def test_unet2d_real_width_vs_oracle(trees):
    from oracle import unet_ref as U
    torch.manual_seed(1)
    net = trees['Unet2D'](dim=128, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x, t = torch.randn(1, 9, 32, 32), torch.tensor([77])
    with torch.no_grad():
        out_ref = U.unet2d_forward(sd, x, t, dim=128, dim_mults=(1, 2, 4, 8), groups=1)
        out = net.to(DEV)(x.to(DEV), t.to(DEV))
    assert rel_l2(out, out_ref) < TOL
