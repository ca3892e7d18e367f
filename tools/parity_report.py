"""Parity report (GPU box): how far is the HIP product path from the reference, measured honestly.

Three evaluations of the SAME function on the SAME inputs are compared:
  hip  : the product path (wdno_amd modules on the MI355X), default arithmetic and WDNO_CONV_MATH=fp32
  cpu32: the CPU oracle in fp32 (on the golden chains this is bit-identical to the reference's own outputs: same torch CPU kernels)
  cpu64: the CPU oracle evaluated in fp64 on the same fp32 weights / inputs = the exact value of the function
so that `hip vs cpu64` is the true error of the product path and `cpu32 vs cpu64` is the round-off the reference's own
fp32 evaluation carries. Prints one JSON document (also written to gpurun_out/parity_report.json).

    python tools/parity_report.py [--quick]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wdno_amd import tree_path  # noqa: E402
for _t in ('third_party', 'smoke', 'burgers'):
    sys.path.insert(0, tree_path(_t))
from oracle import diffusion_ref as D, unet_ref as U  # noqa: E402  (checker only)
from tests.helpers import load_npz, manifest, noise_seq, rel_l2, weights  # noqa: E402
from wdno_amd import ops  # noqa: E402

DEV = 'cuda'
M = manifest()
f64 = torch.float64


def cast_sd(sd, dt):
    return {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}


def cast_buf(buf, dt):
    return {k: v.to(dt) for k, v in buf.items()}


def set_math(mode):
    ops.CONV_MATH = mode


# ------------------------------------------------------------------------------------------------ golden chains (tiny models)
def smoke_chains():
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion
    g = load_npz('ref_smoke_diffusion.npz')
    c = M['smoke_diffusion']
    u, d = c['unet'], dict(c['diffusion'])
    d['padded_shape'] = tuple(d['padded_shape']); d['ori_shape'] = tuple(d['ori_shape'])
    out = {}

    def cpu(dt):
        sd = cast_sd(weights(g, 'w::model.'), dt)
        model = lambda x, t: U.unet3d_forward(sd, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_groups'])
        kw = dict(padded_shape=d['padded_shape'], init=torch.from_numpy(g['ddim_init']).to(dt), control=torch.from_numpy(g['ddim_control']).to(dt))
        with torch.no_grad():
            a = D.smoke_ddim_sample(model, cast_buf(D.make_buffers('sigmoid', 1000), dt), [n.to(dt) for n in noise_seq(g, 'ddim')], 1000, 4, 1.0, **kw)
            b = D.smoke_p_sample_loop(model, cast_buf(D.make_buffers('sigmoid', 5), dt), [n.to(dt) for n in noise_seq(g, 'ddpm5')], 5, **kw)
        return a, b

    def hip():
        net = Unet3D_with_Conv3D(dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_groups=u['resnet_groups'])
        dif = GaussianDiffusion(net, loss_layer_weight=torch.from_numpy(g['lw']), **d)
        dif.load_state_dict(weights(g, 'w::'), strict=True)
        dif = dif.to(DEV)
        init, control = torch.from_numpy(g['ddim_init']).to(DEV), torch.from_numpy(g['ddim_control']).to(DEV)
        seq = iter([n.to(DEV) for n in noise_seq(g, 'ddim')])
        dif.sample_noise = lambda shape, device: next(seq)
        a = dif.sample(batch_size=2, init=init, control=control)
        dif5 = GaussianDiffusion(dif.model, loss_layer_weight=torch.from_numpy(g['lw']), **{**d, 'timesteps': 5, 'sampling_timesteps': None}).to(DEV)
        seq5 = iter([n.to(DEV) for n in noise_seq(g, 'ddpm5')])
        dif5.sample_noise = lambda shape, device: next(seq5)
        b = dif5.sample(batch_size=2, init=init, control=control)
        return a.cpu(), b.cpu()

    a64, b64 = cpu(f64)
    ah, bh = hip()
    for tag, h, e, ref in (('ddim4', ah, a64, g['ddim_out']), ('ddpm5', bh, b64, g['ddpm5_out'])):
        out[tag] = {'hip_vs_reference': rel_l2(h, ref), 'hip_vs_exact': rel_l2(h, e), 'reference_vs_exact': rel_l2(torch.from_numpy(ref), e)}
    return out


def burgers_chains():
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion
    g = load_npz('ref_burgers_diffusion.npz')
    c = M['burgers_diffusion']
    u, d = c['unet'], dict(c['diffusion'])
    d['seq_length'] = tuple(d['seq_length'])
    flags = dict(pad=True, u0=True, uT=False, f=True)
    out = {}

    def cpu(dt):
        sd = cast_sd(weights(g, 'w::model.'), dt)
        model = lambda x, t: U.unet2d_forward(sd, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_block_groups'])
        kw = dict(padded_shape=d['padded_shape'], flags=flags, u0=torch.from_numpy(g['ddim_u_init']).to(dt), f=torch.from_numpy(g['ddim_f']).to(dt))
        with torch.no_grad():
            a = D.burgers_ddim_sample(model, cast_buf(D.make_buffers('cosine', 1000), dt), [n.to(dt) for n in noise_seq(g, 'ddim')], 1000, 4, 1.0, **kw)
            b = D.burgers_p_sample_loop(model, cast_buf(D.make_buffers('cosine', 5), dt), [n.to(dt) for n in noise_seq(g, 'ddpm5')], 5, **kw)
        return a, b

    def make(**over):
        dd = dict(d); dd.update(over)
        net = Unet2D(dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_block_groups=u['resnet_block_groups'])
        dif = GaussianDiffusion(net, loss_layer_weight=torch.from_numpy(g['lw']), **dd)
        sd = weights(g, 'w::')
        if over.get('timesteps'):
            dif.load_state_dict({k: v for k, v in sd.items() if k.startswith('model.')}, strict=False)
        else:
            dif.load_state_dict(sd, strict=True)
        return dif.to(DEV)

    u_init, f = torch.from_numpy(g['ddim_u_init']).to(DEV), torch.from_numpy(g['ddim_f']).to(DEV)
    dif = make()
    seq = iter([n.to(DEV) for n in noise_seq(g, 'ddim')])
    dif.sample_noise = lambda shape, device: next(seq)
    ah = dif.sample(batch_size=2, u_init=u_init, f=f).cpu()
    dif5 = make(timesteps=5, sampling_timesteps=None)
    seq5 = iter([n.to(DEV) for n in noise_seq(g, 'ddpm5')])
    dif5.sample_noise = lambda shape, device: next(seq5)
    bh = dif5.sample(batch_size=2, u_init=u_init, f=f).cpu()
    a64, b64 = cpu(f64)
    for tag, h, e, ref in (('ddim4', ah, a64, g['ddim_out']), ('ddpm5', bh, b64, g['ddpm5_out'])):
        out[tag] = {'hip_vs_reference': rel_l2(h, ref), 'hip_vs_exact': rel_l2(h, e), 'reference_vs_exact': rel_l2(torch.from_numpy(ref), e)}
    return out


# ------------------------------------------------------------------------------------------------ full-size training step
def _grad_table(named_hip, ref32, ref64):
    """worst / median rel-L2 over parameters of: hip vs cpu32, hip vs cpu64, cpu32 vs cpu64."""
    rows = []
    for k, gh in named_hip.items():
        if k not in ref64 or ref64[k] is None:
            continue
        rows.append((k, rel_l2(gh, ref32[k]), rel_l2(gh, ref64[k]), rel_l2(ref32[k], ref64[k])))
    def stat(i):
        v = sorted(r[i] for r in rows)
        return {'worst': v[-1], 'median': v[len(v) // 2], 'worst_param': max(rows, key=lambda r: r[i])[0]}
    return {'n_params': len(rows), 'hip_vs_cpu32': stat(1), 'hip_vs_exact': stat(2), 'cpu32_vs_exact': stat(3)}


def smoke_full(batch=2, frames=24, size=40, modes=('f16x3', 'f32'), channels=42):
    """channels: 42 = the reference's five fields (8 sub-bands each + initial density + smoke-out); 34 = BASELINE's four synthetic fields
    (smoke/ddpm/utils.py:62-63 scaled to 4 fields: 4 x 8 + 2). The conditioning predicates keep the reference's channel positions (C - 2, C - 1, 24:40
    clipped to C), in the oracle and in csrc/diffusion.hip alike."""
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion
    torch.manual_seed(0)
    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=channels)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    lw = torch.linspace(1.0, 22.0, channels).reshape(1, 1, channels, 1, 1)
    ps = (18, 34, 34) if size == 40 else (frames * 3 // 4, size * 3 // 4, size * 3 // 4)
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(batch, frames, channels, size, size, generator=g) * 0.5
    noise = torch.randn(batch, frames, channels, size, size, generator=g)
    t = torch.tensor([37, 911][:batch] if batch <= 2 else list(range(17, 17 + 97 * batch, 97)))
    res = {'shape': list(x0.shape)}

    def cpu(dt):
        sd = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() and not k.endswith('freqs') else (v.to(dt) if v.is_floating_point() else v))
              for k, v in sd0.items()}
        model = lambda x, tt: U.unet3d_forward(sd, x, tt, dim=64, dim_mults=(1, 2, 4), groups=8)
        t0 = time.perf_counter()
        loss = D.smoke_p_losses(model, cast_buf(D.make_buffers('sigmoid', 1000), dt), x0.to(dt), t, noise.to(dt), padded_shape=ps, loss_layer_weight=lw.to(dt))
        loss.backward()
        return loss.item(), {k: v.grad for k, v in sd.items() if v.requires_grad}, time.perf_counter() - t0

    l32, g32, s32 = cpu(torch.float32)
    l64, g64, s64 = cpu(f64)
    res['cpu_seconds'] = {'fp32': round(s32, 2), 'fp64': round(s64, 2)}
    res['loss'] = {'cpu32': l32, 'exact': l64, 'cpu32_vs_exact': abs(l32 - l64) / abs(l64)}
    for mode in modes:
        set_math(mode)
        ops.bump_weight_epoch()
        net_h = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=channels)
        net_h.load_state_dict(sd0)
        dif = GaussianDiffusion(net_h, lw, True, True, True, False, 'bior1.3', 'zero', ps, (32, 64, 64), image_size=size, frames=frames).to(DEV)
        ops.PROFILE = {}
        loss = dif.p_losses(x0.to(DEV), t.to(DEV), noise=noise.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        kernels, ops.PROFILE = sorted(ops.PROFILE), None
        gh = {k: p.grad.cpu() for k, p in net_h.named_parameters() if p.grad is not None}
        res[mode] = {'loss': loss.item(), 'loss_vs_cpu32': abs(loss.item() - l32) / abs(l32), 'loss_vs_exact': abs(loss.item() - l64) / abs(l64),
                     'conv_kernels_used': kernels, 'grads': _grad_table(gh, g32, g64)}
    set_math('f16x3')
    return res


def burgers_full(batch=2, modes=('f16x3', 'f32'), grid=(64, 64), coef=(41, 60), from_fields=False):
    """grid / coef: tensor size and the coefficient region inside it. from_fields: x0 is produced ON THE GPU path from a field batch
    [B, 2, 2 grid] through the HIP DWT (bior2.4 / periodization) + packing (+ one condition channel), the north-star's synthetic Burgers
    input ([B, 2, 160, 128] -> [B, 8 + 1, 80, 64]); the oracle then starts from that same packed tensor."""
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion
    torch.manual_seed(1)
    net = Unet2D(dim=128, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    gh, gw = grid
    if from_fields:
        from wdno_amd import wavelets as Wv
        fields = torch.randn(batch, 2, 2 * gh, 2 * gw, generator=g)
        packed = Wv.dwt_packed(fields.to(DEV), 'bior2.4', 'periodization', 2)            # [B, 2, 4, gh, gw] in coef_to_tensor order
        cond = torch.randn(batch, 1, gh, gw, generator=g) * 0.5
        x0 = torch.cat((packed.reshape(batch, 8, gh, gw).cpu() * 0.25, cond), dim=1).contiguous()
    else:
        x0 = torch.randn(batch, 9, gh, gw, generator=g) * 0.5
    noise = torch.randn(batch, 9, gh, gw, generator=g)
    t = torch.tensor([77, 805][:batch])
    lw = torch.ones(1, 9, 1, 1)
    flags = dict(pad=True, u0=True, uT=False, f=True)
    res = {'shape': list(x0.shape)}

    def cpu(dt):
        sd = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd0.items()}
        model = lambda x, tt: U.unet2d_forward(sd, x, tt, dim=128, dim_mults=(1, 2, 4, 8), groups=1)
        t0 = time.perf_counter()
        loss = D.burgers_p_losses(model, cast_buf(D.make_buffers('cosine', 1000), dt), x0.to(dt), t, noise.to(dt), padded_shape=list(coef),
                                  loss_layer_weight=lw.to(dt), flags=flags)
        loss.backward()
        return loss.item(), {k: v.grad for k, v in sd.items() if v.requires_grad}, time.perf_counter() - t0

    l32, g32, s32 = cpu(torch.float32)
    l64, g64, s64 = cpu(f64)
    res['cpu_seconds'] = {'fp32': round(s32, 2), 'fp64': round(s64, 2)}
    res['loss'] = {'cpu32': l32, 'exact': l64, 'cpu32_vs_exact': abs(l32 - l64) / abs(l64)}
    for mode in modes:
        set_math(mode)
        ops.bump_weight_epoch()
        net_h = Unet2D(dim=128, dim_mults=(1, 2, 4, 8), channels=9, resnet_block_groups=1)
        net_h.load_state_dict(sd0)
        dif = GaussianDiffusion(net_h, seq_length=tuple(grid), padded_shape=list(coef), ori_shape=[2 * coef[0] - 1, 2 * coef[1]], loss_layer_weight=lw,
                                is_condition_pad=True, is_condition_u0=True, is_condition_f=True).to(DEV)
        ops.PROFILE = {}
        loss = dif.p_losses(x0.to(DEV), t.to(DEV), noise=noise.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        kernels, ops.PROFILE = sorted(ops.PROFILE), None
        gh = {k: p.grad.cpu() for k, p in net_h.named_parameters() if p.grad is not None}
        res[mode] = {'loss': loss.item(), 'loss_vs_cpu32': abs(loss.item() - l32) / abs(l32), 'loss_vs_exact': abs(loss.item() - l64) / abs(l64),
                     'conv_kernels_used': kernels, 'grads': _grad_table(gh, g32, g64)}
    set_math('f16x3')
    return res


# ------------------------------------------------------------------------------------------------ full-size DDIM chain A/B
def smoke_chain_full(steps=10, batch=1, modes=('f16x3', 'f32'), seed=7):
    """A `steps`-step DDIM chain (eta = 1, injected noise) of the full-width smoke model at the bench shape."""
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion
    torch.manual_seed(0)
    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    lw = torch.ones(1, 1, 42, 1, 1)
    g = torch.Generator().manual_seed(seed)
    shape = (batch, 24, 42, 40, 40)
    ns = [torch.randn(shape, generator=g) for _ in range(steps + 1)]
    init = torch.randn(batch, 24, 40, 40, generator=g) * 0.3
    control = torch.randn(batch, 24, 16, 40, 40, generator=g) * 0.3
    res = {'shape': list(shape), 'ddim_steps': steps}

    def cpu(dt):
        sd = cast_sd(sd0, dt)
        model = lambda x, tt: U.unet3d_forward(sd, x, tt, dim=64, dim_mults=(1, 2, 4), groups=8)
        t0 = time.perf_counter()
        with torch.no_grad():
            o = D.smoke_ddim_sample(model, cast_buf(D.make_buffers('sigmoid', 1000), dt), [n.to(dt) for n in ns], 1000, steps, 1.0,
                                    padded_shape=(18, 34, 34), init=init.to(dt), control=control.to(dt))
        return o, time.perf_counter() - t0

    o32, s32 = cpu(torch.float32)
    o64, s64 = cpu(f64)
    res['cpu_seconds'] = {'fp32': round(s32, 2), 'fp64': round(s64, 2)}
    res['cpu32_vs_exact'] = rel_l2(o32, o64)
    for mode in modes:
        set_math(mode)
        ops.bump_weight_epoch()
        net_h = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
        net_h.load_state_dict(sd0)
        dif = GaussianDiffusion(net_h, lw, True, True, True, False, 'bior1.3', 'zero', (18, 34, 34), (32, 64, 64), image_size=40, frames=24,
                                sampling_timesteps=steps, ddim_sampling_eta=1.0).to(DEV)
        seq = iter([n.to(DEV) for n in ns])
        dif.sample_noise = lambda shape, device: next(seq)
        o = dif.sample(batch_size=batch, init=init.to(DEV), control=control.to(DEV)).cpu()
        res[mode] = {'hip_vs_cpu32': rel_l2(o, o32), 'hip_vs_exact': rel_l2(o, o64)}
    set_math('f16x3')
    return res


# ------------------------------------------------------------------------------------------------ configs[4] at full size
def sr_full(batch=1, channels=82, train_only=False):
    """BASELINE configs[4] tensor [B, 48, 82, 80, 80] through the space-SR model Unet3D_with_Conv3D(dim=64, (1,2,4), channels=channels):
    one p_losses (loss + every parameter gradient) and one DDIM step replayed from a captured HIP graph, against the fp32 oracle on
    the host (the fp64 evaluation of this size takes minutes and is not run)."""
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion
    from wdno_amd import diffusion_core as K
    avail = 0
    with open('/proc/meminfo') as fh:
        for line in fh:
            if line.startswith('MemAvailable'):
                avail = int(line.split()[1]) / 1e6          # GB
    if avail < 96:          # the fp32 oracle's autograd tape at this size holds ~25 GB; never drive the host out of memory
        return {'skipped': f'host has {avail:.0f} GB available, the full-size oracle step wants >= 96 GB'}
    torch.manual_seed(3)
    shapes = [[18, 34, 34], [34, 66, 66]]
    net = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=channels)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    lw = torch.ones(1, 1, channels, 1, 1)
    g = torch.Generator().manual_seed(9)
    shape = (batch, 48, channels, 80, 80)
    x0 = torch.randn(shape, generator=g) * 0.5
    noise = torch.randn(shape, generator=g)
    t = torch.tensor([423] * batch)
    coef_shape = [shapes[1][0], shapes[1][1] + 2, shapes[1][2] + 2]          # space SR: +2 border coefficients in h and w (diffusion_2d.py:860)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith('freqs') else v) for k, v in sd0.items()}
    model = lambda x, tt: U.unet3d_forward(sd, x, tt, dim=64, dim_mults=(1, 2, 4), groups=8)
    buf = D.make_buffers('sigmoid', 1000)
    t0 = time.perf_counter()
    loss = D.smoke_p_losses(model, buf, x0, t, noise, padded_shape=coef_shape, loss_layer_weight=lw, is_super_model=True)
    loss.backward()
    l32, g32 = loss.item(), {k: v.grad for k, v in sd.items() if v.requires_grad}
    res = {'shape': list(shape), 'cpu_seconds_train_step': round(time.perf_counter() - t0, 1)}
    dif = GaussianDiffusion(net, lw, True, True, True, True, 'bior1.3', 'zero', shapes, [[32, 64, 64], [64, 128, 128]], image_size=80, frames=48,
                            timesteps=1000, sampling_timesteps=10, ddim_sampling_eta=1.0).to(DEV)
    ops.PROFILE = {}
    lh = dif.p_losses(x0.to(DEV), t.to(DEV), noise=noise.to(DEV))
    lh.backward()
    torch.cuda.synchronize()
    kernels, ops.PROFILE = sorted(ops.PROFILE), None
    gh = {k: p.grad.cpu() for k, p in net.named_parameters() if p.grad is not None}
    rows = sorted((rel_l2(gh[k], g32[k]), k) for k in gh if k in g32 and g32[k] is not None)
    res['train'] = {'loss': lh.item(), 'loss_cpu32': l32, 'loss_vs_cpu32': abs(lh.item() - l32) / abs(l32), 'n_params': len(rows), 'worst_grad_vs_cpu32': rows[-1][0],
                    'worst_param': rows[-1][1], 'median_grad_vs_cpu32': rows[len(rows) // 2][0], 'conv_kernels_used': kernels}
    if train_only:
        return res
    # ---- one DDIM step (t = 500 -> 400, eta = 1) replayed from the captured graph
    xs = torch.randn(shape, generator=g)
    nz = torch.randn(shape, generator=g)
    low, init, control = torch.randn(batch, 48, 40, 80, 80, generator=g) * 0.3, torch.randn(batch, 48, 80, 80, generator=g), torch.randn(batch, 48, 16, 80, 80, generator=g)
    kw = dict(init=init, control=control, low=low, is_super_model=True)
    D.smoke_apply_conditions(xs, coef_shape, **kw)
    sdn = {k: v.detach() for k, v in sd.items()}
    with torch.no_grad():
        tt = torch.full((batch,), 500, dtype=torch.long)
        e32, s32 = D.smoke_model_predictions(lambda x, q: U.unet3d_forward(sdn, x, q, dim=64, dim_mults=(1, 2, 4), groups=8), buf, xs, tt, clip_x_start=True, rederive=True)
        n32 = D.smoke_apply_conditions(D.ddim_update(buf, s32, e32, 500, 400, 1.0, nz), coef_shape, **kw)
        desc = dif._desc(shape, dif._coef_shape(shape, 1))
        src = dif._condition_source(shape, DEV, init.to(DEV), control.to(DEV), low.to(DEV))
        sg = K._step_graph(dif, shape, desc, True, False, torch.device(DEV))
        for rep in range(2):            # the second replay of the same graph must reproduce the first
            sg.src.copy_(src); sg.x.copy_(xs.to(DEV)); sg.noise.copy_(nz.to(DEV)); sg.t.fill_(500)
            sigma, c, sqrt_an = K.ddim_coefficients(dif._ac_host, 500, 400, 1.0)
            sg.coef.copy_(torch.tensor([sqrt_an, c, sigma]))
            sg.graph.replay()
            torch.cuda.synchronize()
            res[f'ddim_step_replay{rep}'] = {'x_next_vs_cpu32': rel_l2(sg.x, n32), 'x_start_vs_cpu32': rel_l2(sg.x_start, s32)}
        res['replays_bit_equal'] = res['ddim_step_replay0'] == res['ddim_step_replay1']
    K._graph_cache.pop(dif, None)
    return res


# ------------------------------------------------------------------------------------------------ DDIM-from-t=999 chains over seeds
def chain_seeds(modes=('f16x3',), full_seeds=0):
    """is the HIP path systematically further from the exact value of a DDIM chain that starts at t = 999 than the
    reference's own fp32 evaluation is? tests/golden/ref_round3.npz holds, for 8 seeds x {smoke, Burgers} (different weights,
    conditions and noise per seed), the REFERENCE's fp32 DDIM-4 output (eta = 1); here every chain is evaluated by the HIP path and by
    the fp64 oracle (= the exact value of the chain) and three distances are recorded per seed:
        hip_vs_exact, ref_vs_exact (reference fp32 vs exact), hip_vs_ref.
    With full_seeds > 0 the same is done for that many seeds of the full-size 10-step smoke chain (cpu32 oracle in place of the
    reference, which is bit-identical to it on the golden chains)."""
    from ddpm_burgers.unet import Unet2D
    from ddpm_burgers.diffusion_1d import GaussianDiffusion as GD1
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion as GD2
    import json as _json
    from tests.helpers import GOLDEN, randomise
    g = load_npz('ref_round3.npz')
    with open(os.path.join(GOLDEN, 'ref_round3_manifest.json')) as fh:
        meta = _json.load(fh)['mseed']
    base = {tree: {k[len(f'mseed_base::{tree}::'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f'mseed_base::{tree}::')}
            for tree in ('sm', 'bu', 'buf_sm', 'buf_bu')}           # buf_*: the schedule buffers the reference outputs belong to
    out = {'smoke': [], 'burgers': []}
    for s in range(meta['n_seeds']):
        # ---- smoke
        u, d = meta['smoke']['unet'], dict(meta['smoke']['diffusion'])
        d['padded_shape'] = tuple(d['padded_shape']); d['ori_shape'] = tuple(d['ori_shape'])
        sub = {k[len(f'mseed_smoke{s}::'):]: g[k] for k in g.files if k.startswith(f'mseed_smoke{s}::')}
        net = Unet3D_with_Conv3D(dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_groups=u['resnet_groups'],
                                 init_kernel_size=u['init_kernel_size'])
        net.load_state_dict(base['sm'])
        randomise(net, torch.Generator().manual_seed(1000 + s))
        sd32 = {k: v.detach().clone() for k, v in net.state_dict().items()}
        ns = [torch.from_numpy(sub[f'ddim_noise_{i}']) for i in range(int(sub['ddim_n_noise']))]
        init, control = torch.from_numpy(sub['init']), torch.from_numpy(sub['control'])
        sd64 = cast_sd(sd32, f64)
        with torch.no_grad():
            m64 = lambda x, t: U.unet3d_forward(sd64, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_groups'])
            e = D.smoke_ddim_sample(m64, cast_buf(base['buf_sm'], f64), [n.double() for n in ns], 1000, d['sampling_timesteps'], 1.0,
                                    padded_shape=d['padded_shape'], init=init.double(), control=control.double())
        row = {'seed': s, 'ref_vs_exact': rel_l2(torch.from_numpy(sub['ddim_out']), e)}
        for mode in modes:
            set_math(mode)
            ops.bump_weight_epoch()
            dif = GD2(net, loss_layer_weight=torch.ones(1, 1, u['channels'], 1, 1), **d)
            own = {k: v.clone() for k, v in dif.state_dict().items() if k in base['buf_sm']}
            dif.load_state_dict(base['buf_sm'], strict=False)
            dif = dif.to(DEV)
            seq = iter([n.to(DEV) for n in ns])
            dif.sample_noise = lambda shape, device: next(seq)
            h = dif.sample(batch_size=2, init=init.to(DEV), control=control.to(DEV)).cpu()
            row[mode] = {'hip_vs_exact': rel_l2(h, e), 'hip_vs_ref': rel_l2(h, sub['ddim_out'])}
        out['smoke'].append(row)
        # ---- Burgers
        u, d = meta['burgers']['unet'], dict(meta['burgers']['diffusion'])
        d['seq_length'] = tuple(d['seq_length'])
        sub = {k[len(f'mseed_burgers{s}::'):]: g[k] for k in g.files if k.startswith(f'mseed_burgers{s}::')}
        net = Unet2D(dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_block_groups=u['resnet_block_groups'])
        net.load_state_dict(base['bu'])
        randomise(net, torch.Generator().manual_seed(2000 + s))
        if 'w_check' in sub:
            assert torch.equal(net.init_conv.weight.detach(), torch.from_numpy(sub['w_check'])), 'per-seed weights were not re-created bit-exactly'
        sd32 = {k: v.detach().clone() for k, v in net.state_dict().items()}
        ns = [torch.from_numpy(sub[f'ddim_noise_{i}']) for i in range(int(sub['ddim_n_noise']))]
        u0, ff = torch.from_numpy(sub['u_init']), torch.from_numpy(sub['f'])
        sd64 = cast_sd(sd32, f64)
        flags = dict(pad=True, u0=True, uT=False, f=True)
        with torch.no_grad():
            m64 = lambda x, t: U.unet2d_forward(sd64, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_block_groups'])
            e = D.burgers_ddim_sample(m64, cast_buf(base['buf_bu'], f64), [n.double() for n in ns], 1000, d['sampling_timesteps'], 1.0,
                                      padded_shape=d['padded_shape'], flags=flags, u0=u0.double(), f=ff.double())
        row = {'seed': s, 'ref_vs_exact': rel_l2(torch.from_numpy(sub['ddim_out']), e)}
        for mode in modes:
            set_math(mode)
            ops.bump_weight_epoch()
            dif = GD1(net, loss_layer_weight=torch.ones(1, u['channels'], 1, 1), **d)
            own_b = {k: v.clone() for k, v in dif.state_dict().items() if k in base['buf_bu']}
            dif.load_state_dict(base['buf_bu'], strict=False)
            dif = dif.to(DEV)
            seq = iter([n.to(DEV) for n in ns])
            dif.sample_noise = lambda shape, device: next(seq)
            h = dif.sample(batch_size=2, u_init=u0.to(DEV), f=ff.to(DEV)).cpu()
            row[mode] = {'hip_vs_exact': rel_l2(h, e), 'hip_vs_ref': rel_l2(h, sub['ddim_out'])}
        out['burgers'].append(row)
    set_math('f16x3')
    # how many schedule entries does THIS host compute differently (last fp32 bit) from the host the fixture was made on?
    out['schedule_entries_differing_from_fixture_host'] = {
        'smoke_sigmoid': {k: int((own[k].cpu() != base['buf_sm'][k]).sum()) for k in own},
        'burgers_cosine': {k: int((own_b[k].cpu() != base['buf_bu'][k]).sum()) for k in own_b}}
    med = lambda v: sorted(v)[len(v) // 2]
    for tree in ('smoke', 'burgers'):
        rows = out[tree]
        summ = {'median_ref_vs_exact': med([r['ref_vs_exact'] for r in rows]), 'max_ref_vs_exact': max(r['ref_vs_exact'] for r in rows)}
        for mode in modes:
            summ[mode] = {'median_hip_vs_exact': med([r[mode]['hip_vs_exact'] for r in rows]), 'max_hip_vs_exact': max(r[mode]['hip_vs_exact'] for r in rows),
                          'median_hip_vs_ref': med([r[mode]['hip_vs_ref'] for r in rows]), 'max_hip_vs_ref': max(r[mode]['hip_vs_ref'] for r in rows),
                          'seeds_hip_closer_to_exact_than_ref': sum(1 for r in rows if r[mode]['hip_vs_exact'] <= r['ref_vs_exact'])}
        out[tree + '_summary'] = summ
    if full_seeds:
        out['smoke_full_ddim10'] = [dict(seed=s, **{k: v for k, v in smoke_chain_full(10, 1, modes=('f16x3',), seed=7 + s).items()
                                                    if k in ('cpu32_vs_exact', 'f16x3')}) for s in range(full_seeds)]
    return out


# ------------------------------------------------------------------------------------------------ per-step errors from a common state
def smoke_chain_steps(steps=4):
    """Every DDIM step of the golden smoke configuration, started from the SAME state (the fp64 chain's state rounded to fp32) and
    evaluated by the HIP path, the fp32 CPU oracle and the fp64 oracle. Separates what a step adds (U-Net round-off, and at
    t = T-1 the fp32 arithmetic of x_start = c1 x - c2 eps with c1 ~ c2 ~ 1.8e3) from what the chain amplifies."""
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion
    from wdno_amd import diffusion_core as K
    g = load_npz('ref_smoke_diffusion.npz')
    c = M['smoke_diffusion']
    u, d = c['unet'], dict(c['diffusion'])
    d['padded_shape'] = tuple(d['padded_shape']); d['ori_shape'] = tuple(d['ori_shape'])
    d['sampling_timesteps'] = steps
    net = Unet3D_with_Conv3D(dim=u['dim'], dim_mults=tuple(u['dim_mults']), channels=u['channels'], resnet_groups=u['resnet_groups'])
    dif = GaussianDiffusion(net, loss_layer_weight=torch.from_numpy(g['lw']), **d)
    dif.load_state_dict(weights(g, 'w::'), strict=True)
    dif = dif.to(DEV)
    sd32 = weights(g, 'w::model.')
    sd64 = cast_sd(sd32, f64)
    cfg = dict(dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_groups'])
    b32 = D.make_buffers('sigmoid', 1000)
    b64 = cast_buf(b32, f64)
    gen = torch.Generator().manual_seed(7)
    shape = (2, 4, 42, 8, 8)
    ns = [torch.randn(shape, generator=gen) for _ in range(steps + 1)]
    init, control = torch.from_numpy(g['ddim_init']), torch.from_numpy(g['ddim_control'])
    x = ns[0].double().clone()
    D.smoke_apply_conditions(x, d['padded_shape'], init=init.double(), control=control.double())
    desc = dif._desc(shape, dif.padded_shape)
    src = dif._condition_source(shape, DEV, init.to(DEV), control.to(DEV), None)
    rows = []
    with torch.no_grad():
        for i, (time, time_next) in enumerate(D.ddim_times(1000, steps)):
            t = torch.full((2,), time, dtype=torch.long)
            x32 = x.float()
            m64 = lambda a, b: U.unet3d_forward(sd64, a, b, **cfg)
            m32 = lambda a, b: U.unet3d_forward(sd32, a, b, **cfg)
            raw64, raw32 = m64(x32.double(), t), m32(x32, t)
            e64, s64 = D.smoke_model_predictions(m64, b64, x32.double(), t, clip_x_start=True, rederive=True)
            e32, s32 = D.smoke_model_predictions(m32, b32, x32, t, clip_x_start=True, rederive=True)
            rawh = dif.model(x32.to(DEV), t.to(DEV), None)
            if time_next < 0:
                n64, n32 = s64, s32
                nh, sh = K.ddim_update(dif, x32.to(DEV), rawh, None, t.to(DEV), 0., 0., 0.)
            else:
                n64 = D.ddim_update(b64, s64, e64, time, time_next, 1.0, ns[i + 1].double())
                n32 = D.ddim_update(b32, s32, e32, time, time_next, 1.0, ns[i + 1])
                sigma, cc, sqrt_an = K.ddim_coefficients(dif._ac_host, time, time_next, 1.0)
                nh, sh = K.ddim_update(dif, x32.to(DEV), rawh, ns[i + 1].to(DEV), t.to(DEV), sqrt_an, cc, sigma)
                D.smoke_apply_conditions(n64, d['padded_shape'], init=init.double(), control=control.double())
                D.smoke_apply_conditions(n32, d['padded_shape'], init=init, control=control)
                nh = K.apply_cond(nh, src, desc)
            rows.append({'t': time, 't_next': time_next, 'c2': float(b32['sqrt_recipm1_alphas_cumprod'][time]),
                         'eps': {'hip': rel_l2(rawh, raw64), 'cpu32': rel_l2(raw32, raw64)},
                         'x_start': {'hip': rel_l2(sh, s64), 'cpu32': rel_l2(s32, s64)},
                         'x_next': {'hip': rel_l2(nh, n64), 'cpu32': rel_l2(n32, n64)}})
            x = n64
    return {'steps': rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--h3-min-pixels', type=int, default=None, help='experiment: override ops.H3_MIN_PIXELS')
    ap.add_argument('--h3-min-reduction', type=int, default=None)
    ap.add_argument('--only', default=None, help='comma-separated section names')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'parity_report.json'))
    args = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    if args.h3_min_pixels is not None:
        ops.H3_MIN_PIXELS = args.h3_min_pixels
    if args.h3_min_reduction is not None:
        ops.H3_MIN_REDUCTION = args.h3_min_reduction
    rep = {'settings': {'H3_MIN_PIXELS': ops.H3_MIN_PIXELS, 'H3_MIN_REDUCTION': ops.H3_MIN_REDUCTION}}
    for name, fn in (('smoke_golden_chains', smoke_chains), ('burgers_golden_chains', burgers_chains),
                     ('smoke_full_train_step', smoke_full), ('burgers_full_train_step', burgers_full),
                     ('smoke_full_ddim_chain', (lambda: smoke_chain_full(4)) if args.quick else smoke_chain_full),
                     ('smoke_golden_chain_steps', smoke_chain_steps), ('ddim_chain_seeds', lambda: chain_seeds(('f16x3', 'f32'), 0 if args.quick else 3)),
                     ('burgers_north_star_train_step', lambda: burgers_full(2, ('f16x3',), (80, 64), (80, 64), True)), ('sr_full_size', sr_full)):
        if args.only and name not in args.only.split(','):
            continue
        t0 = time.perf_counter()
        try:
            rep[name] = fn()
        except Exception as e:      # keep going: the report is diagnostic
            import traceback
            rep[name] = {'error': repr(e), 'trace': traceback.format_exc()[-1500:]}
        rep[name]['seconds'] = round(time.perf_counter() - t0, 1)
        print(name, json.dumps(rep[name], indent=1), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump(rep, f, indent=1)


if __name__ == '__main__':
    main()
