"""CPU evaluations of reference chains by the oracle, in fp32 (what the reference computes: pinned against the reference-generated
fixtures by tests/test_oracle_arbiter.py) and in fp64 on the same fp32 weights / inputs / schedule constants (the exact value of the
chain: the ARBITER of the GPU gates). A gate of the form

    hip_vs_exact <= 1.5 * reference_vs_exact + 1e-6

says: the HIP result is no further from the exact chain than (1.5 x) the reference's own fp32 evaluation is. Test infrastructure only."""
import json
import os

import numpy as np
import torch

from oracle import diffusion_ref as D, dwt_ref as R, unet_ref as U
from tests.helpers import GOLDEN

_G2 = None


def round2():
    global _G2
    if _G2 is None:
        with open(os.path.join(GOLDEN, 'ref_round2_manifest.json')) as f:
            _G2 = (np.load(os.path.join(GOLDEN, 'ref_round2.npz')), json.load(f))
    return _G2


def sub(tag):
    g, _ = round2()
    pre = tag + '::'
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


def _sd(gz, prefix, dt):
    out = {k[len(prefix):]: torch.from_numpy(v) for k, v in gz.items() if k.startswith(prefix)}
    return {k: (v.to(dt) if v.is_floating_point() else v) for k, v in out.items()}


def _buf(kind, T, dt):
    return {k: v.to(dt) for k, v in D.make_buffers(kind, T).items()}       # the reference's fp32 schedule constants, exactly


def _noises(gz, tag, dt):
    return [torch.from_numpy(gz[f'{tag}_noise_{i}']).to(dt) for i in range(int(gz[f'{tag}_n_noise']))]


def gate(hip_vs_exact, ref_vs_exact, factor=1.5, slack=1e-6):
    return hip_vs_exact <= factor * ref_vs_exact + slack


def gate_or_bar(hip_vs_exact, ref_vs_exact, bar=1e-5):
    """The north-star bar met outright against the exact value, or the arbiter ratio."""
    return hip_vs_exact < bar or gate(hip_vs_exact, ref_vs_exact)


# --------------------------------------------------------------------------------------------- guided smoke chains (fixture 'gs')
def smoke_guided(dt):
    """sample(design_fn=..., design_guidance='standard') as DDIM-4 from t = 999 and ('standard-alpha') as 6 ancestral steps
    (smoke/inference_2d.py:30-66 -> diffusion_2d.py:723-754,851-933). Returns {'ddim': x, 'ddpm6': x}."""
    gz, meta = sub('gs'), round2()[1]['gs']
    d, u = meta['diffusion'], meta['unet']
    sd = _sd(gz, 'w::model.', dt)
    target, wgt, init, init_u = (torch.from_numpy(gz[k]).to(dt) for k in ('target', 'wgt', 'init', 'init_u'))

    def design_fn(x, low=None, init=None, init_u=None):
        J = 0.5 * (wgt * (x - target) ** 2).sum() + 0.1 * ((x[:, 0, 0] - init_u) ** 2).sum()
        return torch.autograd.grad(J, x)[0]
    model = lambda x, t: U.unet3d_forward(sd, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_groups'])
    gd = dict(design_fn=design_fn, standard_fixed_ratio=d['standard_fixed_ratio'], coeff_ratio=d['coeff_ratio'], init=init, init_u=init_u)
    kw = dict(padded_shape=tuple(d['padded_shape']), init=init, is_condition_control=False)
    with torch.no_grad():
        a = D.smoke_ddim_sample(model, _buf('sigmoid', 1000, dt), _noises(gz, 'ddim', dt), 1000, 4, 1.0, guidance=dict(gd, kind='standard'), **kw)
        b = D.smoke_p_sample_loop(model, _buf('sigmoid', 6, dt), _noises(gz, 'ddpm6', dt), 6, guidance=dict(gd, kind='standard-alpha'), **kw)
    return {'ddim': a, 'ddpm6': b}


# --------------------------------------------------------------------------------------------- super-resolution cascade (fixture 'sr')
BANDS = ('aad', 'ada', 'add', 'daa', 'dad', 'dda', 'ddd')


def _repack(wave, shp, up_type=None):
    """tensor_to_coef -> coef_to_tensor -> [N, T, 40, H, W] (inference_2d.py:166-172), numpy, any dtype."""
    x = wave[:, :, :40].permute(0, 2, 1, 3, 4).numpy()
    lll, det = R.smoke_tensor_to_coef(x, shp, upsample_type=up_type)
    t = R.smoke_coef_to_tensor(lll, det)
    ret = t.reshape(-1, 5, 8, *shp).reshape(-1, 40, *shp).transpose(0, 2, 1, 3, 4)
    return (lll, det), torch.from_numpy(np.ascontiguousarray(ret))


def smoke_cascade(dt, wave0=None):
    """inference_2d.py:155-215 at the fixture's size: base DDIM-3 -> repack -> space up-sampling -> padded `low` -> SR DDIM-3 ->
    coefficients -> IDWT. wave0: start the SR stage from this base sample instead of the oracle's own."""
    gz, m = sub('sr'), round2()[1]['sr']
    shape, pad_t, (pad_x0, pad_x1) = m['shape'], m['pad_t'], m['pad_x']
    t = lambda k: torch.from_numpy(gz[k]).to(dt)
    mk = lambda sd: (lambda x, tt: U.unet3d_forward(sd, x, tt, dim=8, dim_mults=(1, 2), groups=4))
    buf = _buf('sigmoid', 1000, dt)
    out = {}
    with torch.no_grad():
        if wave0 is None:
            wave0 = D.smoke_ddim_sample(mk(_sd(gz, 'w0::model.', dt)), buf, _noises(gz, 'base', dt), 1000, 3, 1.0, padded_shape=tuple(shape[0]),
                                        init=t('init0'), control=t('control0'))
        out['wave0'] = wave0
        _, ret0 = _repack(wave0, shape[0])
        up = torch.from_numpy(R.upsample_coef_3d(ret0.numpy(), 'space'))
        low = torch.nn.functional.pad(up, (0, pad_x1 - up.shape[-1], 0, pad_x1 - up.shape[-2], 0, 0, 0, pad_t - up.shape[-4]), 'constant', 0)
        out['ret0'], out['up'], out['low'] = ret0, up, low
        cs1 = (shape[1][0], shape[1][1] + 2, shape[1][2] + 2)           # diffusion_2d.py:857-860 (is_condition_control)
        wave1 = D.smoke_ddim_sample(mk(_sd(gz, 'w1::model.', dt)), buf, _noises(gz, 'sr', dt), 1000, 3, 1.0, padded_shape=cs1,
                                    init=t('init1'), control=t('control1'), low=low, is_super_model=True)
        out['wave1'] = wave1
        (lll, det), _ = _repack(wave1, shape[1], 'space')
        out['rec'] = torch.from_numpy(R.idwt3(lll.astype(np.float64), {k: v.astype(np.float64) for k, v in det.items()}, 'bior1.3'))
    return out


# --------------------------------------------------------------------------------------------- three optimiser steps (rows T1 / T2)
def _train3(sd0, model_of, loss_of, steps, lr, sched, dt):
    """loss -> backward -> clip_grad_norm_(1.0) -> Adam(lr, (0.9, 0.99)) -> scheduler, `steps` times (train_diffusion.py:203-216,
    diffusion_2d.py:1277-1293) on the oracle's functional model. Returns losses, gradient norms and the final parameters."""
    sd = {k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() and not k.endswith('freqs') else v) for k, v in sd0.items()}
    params = [v for v in sd.values() if torch.is_tensor(v) and v.requires_grad]
    opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.99))
    sch = sched(opt)
    losses, gns = [], []
    for st in steps:
        loss = loss_of(model_of(sd), st)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step(); opt.zero_grad(); sch.step()
        losses.append(float(loss.detach().double())); gns.append(float(gn.double()))
    return losses, gns, {k: v.detach() for k, v in sd.items()}


def burgers_train3(dt):
    from tests.helpers import load_npz, manifest
    gz, c = load_npz('ref_train_burgers.npz'), manifest()['train_burgers']
    u = c['unet']
    sd0 = {k[len('w0::model.'):]: torch.from_numpy(gz[k]) for k in gz.files if k.startswith('w0::model.')}
    buf = _buf('cosine', 1000, dt)
    flags = dict(pad=True, u0=True, uT=False, f=True)
    steps = [tuple(torch.from_numpy(gz[f's{i}_{k}']) for k in ('x0', 't', 'noise')) for i in range(3)]
    model_of = lambda sd: (lambda x, t: U.unet2d_forward(sd, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_block_groups']))
    loss_of = lambda model, st: D.burgers_p_losses(model, buf, st[0].to(dt), st[1], st[2].to(dt), padded_shape=c['padded_shape'],
                                                   loss_layer_weight=torch.ones(1, 9, 1, 1, dtype=dt), flags=flags)
    return _train3(sd0, model_of, loss_of, steps, 1e-4, lambda o: torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=10000), dt)


def smoke_train3(dt):
    gz, m = sub('t2'), round2()[1]['t2']
    u = m['unet']
    sd0 = {k[len('w0::model.'):]: torch.from_numpy(v) for k, v in gz.items() if k.startswith('w0::model.')}
    buf = _buf('sigmoid', 1000, dt)
    lw = torch.from_numpy(gz['lw']).to(dt)
    steps = [tuple(torch.from_numpy(gz[f's{i}_{k}']) for k in ('x0', 't', 'noise')) for i in range(3)]
    model_of = lambda sd: (lambda x, t: U.unet3d_forward(sd, x, t, dim=u['dim'], dim_mults=tuple(u['dim_mults']), groups=u['resnet_groups']))
    loss_of = lambda model, st: D.smoke_p_losses(model, buf, st[0].to(dt), st[1], st[2].to(dt), padded_shape=tuple(m['diffusion']['padded_shape']),
                                                 loss_layer_weight=lw)
    return _train3(sd0, model_of, loss_of, steps, 1e-3,
                   lambda o: torch.optim.lr_scheduler.MultiStepLR(o, milestones=[50000, 150000, 300000], gamma=0.1), dt)


def update_errors(w0, final, exact):
    """rel-L2 of the parameter UPDATES (final - w0) against the exact updates: ({name: err}, err over all moved parameters together)."""
    per, num, den = {}, 0.0, 0.0
    for k, e in exact.items():
        if not (torch.is_tensor(e) and e.is_floating_point()) or k not in final:
            continue
        a, b = final[k].double().cpu() - w0[k].double(), e.double() - w0[k].double()
        if float(b.abs().max()) == 0.0:
            continue
        per[k] = float((a - b).norm() / b.norm())
        num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
    return per, (num / den) ** 0.5


def check_updates(tag, w0, ours, ref, exact):
    """Parameter updates after the optimiser steps: Adam turns every gradient into a step of ~lr whatever its size, so the entries whose
    gradients are round-off-sized move differently in ANY two fp32 evaluations (the reference's own updates are ~1e-4 from the exact ones,
    single parameters up to 5e-4, and which parameter draws the large error differs from evaluation to evaluation). Gates: all moved
    parameters together -- the HIP updates are no further from the exact updates than 1.5 x the reference's; every single parameter -- no
    further than 1.5 x the reference's WORST parameter (the envelope of the same distribution, not its draw on that parameter)."""
    hp, ht = update_errors(w0, ours, exact)
    rp, rt = update_errors(w0, ref, exact)
    rworst = max(rp.values())
    kw = max(hp, key=lambda k: hp[k])
    print(f'{tag}: parameter updates vs exact -- all together: hip {ht:.3e} reference {rt:.3e}; worst parameter: hip {hp[kw]:.3e} ({kw}) '
          f'reference {rworst:.3e}; parameters where hip > 1.5 x reference on the same parameter: '
          f'{sum(1 for k in hp if not gate(hp[k], rp[k]))} of {len(hp)}')
    assert gate(ht, rt), (ht, rt)
    for k in hp:
        assert gate(hp[k], rworst), (k, hp[k], rworst)
