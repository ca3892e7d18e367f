"""ORACLE (test infrastructure only) -- functional CPU restatement of the two GaussianDiffusion operators.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

`model` below is any callable (x, t) -> eps_hat. All randomness is injected (`noise`, `noise_seq`) so that CPU and
GPU runs consume identical values. Pinned against tests/golden/ref_{smoke,burgers}_diffusion.npz and
ref_schedules.npz (outputs of the reference itself).

Follows
  burgers/ddpm_burgers/diffusion_1d.py:97-156,172-258,276-307,310-460,520-645 ; model_utils.py:27-30,141-157
  smoke/ddpm/diffusion_2d.py:508-547,627-685,689-785,788-933,970-1050
"""
import math

import torch


# ----------------------------------------------------------------------------- schedules (fp64 -> fp32 buffers)
def beta_schedule(kind, T):
    if kind == 'linear':            # model_utils.py:141-145 / diffusion_2d.py:513-520
        s = 1000 / T
        return torch.linspace(s * 1e-4, s * 0.02, T, dtype=torch.float64)
    steps = T + 1
    u = torch.linspace(0, T, steps, dtype=torch.float64) / T
    if kind == 'cosine':            # model_utils.py:147-157 / diffusion_2d.py:522-532
        ac = torch.cos((u + 0.008) / 1.008 * math.pi * 0.5) ** 2
    elif kind == 'sigmoid':         # diffusion_2d.py:534-547 (start=-3, end=3, tau=1)
        start, end, tau = -3, 3, 1
        v0 = torch.tensor(start / tau).sigmoid()
        v1 = torch.tensor(end / tau).sigmoid()
        ac = (-((u * (end - start) + start) / tau).sigmoid() + v1) / (v1 - v0)
    else:
        raise ValueError(f'unknown beta schedule {kind}')
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


def make_buffers(kind, T):
    """The 13 registered buffers of either GaussianDiffusion (diffusion_1d.py:115-156, diffusion_2d.py:646-685)."""
    betas = beta_schedule(kind, T)
    alphas = 1. - betas
    ac = torch.cumprod(alphas, dim=0)
    acp = torch.cat([torch.ones(1, dtype=torch.float64), ac[:-1]])
    pv = betas * (1. - acp) / (1. - ac)
    snr = ac / (1 - ac)
    b = dict(
        betas=betas, alphas_cumprod=ac, alphas_cumprod_prev=acp,
        sqrt_alphas_cumprod=ac.sqrt(), sqrt_one_minus_alphas_cumprod=(1. - ac).sqrt(),
        log_one_minus_alphas_cumprod=(1. - ac).log(), sqrt_recip_alphas_cumprod=(1. / ac).sqrt(),
        sqrt_recipm1_alphas_cumprod=(1. / ac - 1).sqrt(), posterior_variance=pv,
        posterior_log_variance_clipped=pv.clamp(min=1e-20).log(),
        posterior_mean_coef1=betas * acp.sqrt() / (1. - ac), posterior_mean_coef2=(1. - acp) * alphas.sqrt() / (1. - ac),
        loss_weight=snr / snr,      # == 1 for pred_noise (Burgers) and for min_snr_loss_weight=False (smoke)
    )
    return {k: v.to(torch.float32) for k, v in b.items()}


def _ex(a, t, x):
    return a.gather(-1, t).reshape(t.shape[0], *([1] * (x.dim() - 1)))


def q_sample(buf, x0, t, noise):
    return _ex(buf['sqrt_alphas_cumprod'], t, x0) * x0 + _ex(buf['sqrt_one_minus_alphas_cumprod'], t, x0) * noise


def predict_start_from_noise(buf, x, t, eps):
    return _ex(buf['sqrt_recip_alphas_cumprod'], t, x) * x - _ex(buf['sqrt_recipm1_alphas_cumprod'], t, x) * eps


def predict_noise_from_start(buf, x, t, x0):
    return (_ex(buf['sqrt_recip_alphas_cumprod'], t, x) * x - x0) / _ex(buf['sqrt_recipm1_alphas_cumprod'], t, x)


def posterior_step(buf, x, t_int, x_start, noise):
    """q_posterior + the p_sample draw (diffusion_1d.py:196-203,251-258 ; diffusion_2d.py:714-721,769-785)."""
    t = torch.full((x.shape[0],), t_int, dtype=torch.long)
    mean = _ex(buf['posterior_mean_coef1'], t, x) * x_start + _ex(buf['posterior_mean_coef2'], t, x) * x
    logvar = _ex(buf['posterior_log_variance_clipped'], t, x)
    if t_int == 0:
        return mean
    return mean + (0.5 * logvar).exp() * noise


def ddim_times(T, S):
    times = list(reversed(torch.linspace(-1, T - 1, steps=S + 1).int().tolist()))
    return list(zip(times[:-1], times[1:]))


def ddim_scalars(buf, time, time_next, eta):
    """sigma, c, sqrt(alpha_next) of diffusion_2d.py:905-909 / diffusion_1d.py:423-428 in the dtype of the schedule buffers. For fp32
    buffers every operation is an IEEE-correctly-rounded fp32 operation (numpy float32 scalars): torch's CPU fp32 sqrt is off by one
    ulp on ~0.5 % of inputs and not on the same ones on every host, and c^2 = 1 - a' - sigma^2 at t = T-1 is ~1e-6 left over from
    terms ~1 (quantised in 4 % steps), so with torch scalars the SAME reference code gives c = 1.196e-3 in the build container and
    1.245e-3 on the GPU box's host, 5e-5 apart at the end of a 4-step chain (tools/ddim_coef_probe.py)."""
    import numpy as np
    ac = buf['alphas_cumprod']
    if ac.dtype == torch.float32:
        one, e = np.float32(1.0), np.float32(eta)
        a, an = np.float32(ac[time].item()), np.float32(ac[time_next].item())
        with np.errstate(invalid='ignore'):
            sigma = e * np.sqrt((one - a / an) * (one - an) / (one - a))
            c = np.sqrt(one - an - sigma * sigma)
            return float(sigma), float(c), float(np.sqrt(an))
    a, an = ac[time], ac[time_next]
    sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
    c = (1 - an - sigma ** 2).sqrt()
    return sigma, c, an.sqrt()


def ddim_update(buf, x_start, eps, time, time_next, eta, noise):
    sigma, c, sqrt_an = ddim_scalars(buf, time, time_next, eta)
    return x_start * sqrt_an + c * eps + sigma * noise


# ----------------------------------------------------------------------------- smoke (x is [B, F, C, H, W])
def smoke_apply_conditions(x, coef_shape, *, init=None, control=None, low=None, is_condition_control=True,
                           is_condition_pad=True, is_super_model=False, zero=False):
    """The in-place overwrites of diffusion_2d.py:1008-1033 (train) and :805-823,:913-929 (sampling), wavelet mode.

    zero=True writes zeros instead of the clean values (what is done to the noise target)."""
    z = (lambda v: torch.zeros_like(v)) if zero else (lambda v: v)
    x[:, :, -2] = z(init)
    if is_condition_control:
        x[:, :, 24:40] = z(control)
    if is_condition_pad:
        x[:, coef_shape[-3]:, :-2] = 0
        x[:, coef_shape[-3]:, -1] = 0
        x[:, :, :-1, coef_shape[-2]:] = 0
        x[:, :, :-1, :, coef_shape[-1]:] = 0
    if is_super_model:
        x[:, :, 40:80] = z(low)
    return x


def smoke_p_losses(model, buf, x0, t, noise, *, padded_shape, loss_layer_weight, is_condition_control=True,
                   is_condition_pad=True, is_super_model=False):
    """diffusion_2d.py:988-1050. Note the loss quirk: scalar mse * weight tensor, then .mean()."""
    noise = noise.clone()
    x = q_sample(buf, x0, t, noise)
    kw = dict(is_condition_control=is_condition_control, is_condition_pad=is_condition_pad, is_super_model=is_super_model)
    smoke_apply_conditions(x, padded_shape, init=x0[:, :, -2], control=x0[:, :, 24:40], low=x0[:, :, 40:80] if is_super_model else None, **kw)
    smoke_apply_conditions(noise, padded_shape, init=noise[:, :, -2], control=noise[:, :, 24:40],
                           low=noise[:, :, 40:80] if is_super_model else None, zero=True, **kw)
    out = model(x, t)
    loss = torch.nn.functional.mse_loss(out, noise, reduction='mean')
    return (loss * loss_layer_weight).mean()


def smoke_model_predictions(model, buf, x, t, clip_x_start=False, rederive=False, guidance=None):
    """diffusion_2d.py:723-754. guidance = None or a dict(design_fn=..., kind='standard' | 'standard-alpha', standard_fixed_ratio=...,
    coeff_ratio=..., low=, init=, init_u=): the callback's gradient at the (clipped) x_start is added to pred_noise -- scaled by
    standard_fixed_ratio, or by coeff_ratio * betas.flip(0)[t] -- BEFORE x_start is derived (and clipped) a second time."""
    eps = model(x, t)
    xs = predict_start_from_noise(buf, x, t, eps)
    if clip_x_start:
        xs = xs.clamp(-1., 1.)
    if guidance is not None:
        with torch.enable_grad():
            xc = xs.clone().detach().requires_grad_()
            g = guidance['design_fn'](xc, low=guidance.get('low'), init=guidance.get('init'), init_u=guidance.get('init_u'))
        g = g.detach()
        if guidance['kind'] == 'standard':
            eps = eps + guidance['standard_fixed_ratio'] * g
        elif guidance['kind'] == 'standard-alpha':
            eps = eps + _ex(guidance['coeff_ratio'] * buf['betas'].flip(0), t, x) * g
        else:
            raise ValueError(guidance['kind'])
        xs = predict_start_from_noise(buf, x, t, eps)
        if clip_x_start:
            xs = xs.clamp(-1., 1.)
    if clip_x_start and rederive:
        eps = predict_noise_from_start(buf, x, t, xs)
    return eps, xs


def smoke_p_sample(model, buf, x, t_int, noise, guidance=None):
    """diffusion_2d.py:757-785 (clip_denoised=True) -> (x_{t-1}, x_start)."""
    t = torch.full((x.shape[0],), t_int, dtype=torch.long)
    _, xs = smoke_model_predictions(model, buf, x, t, guidance=guidance)
    xs = xs.clamp(-1., 1.)
    return posterior_step(buf, x, t_int, xs, noise), xs


def smoke_p_sample_loop(model, buf, noise_seq, T, *, padded_shape, init, control=None, guidance=None, **kw):
    """diffusion_2d.py:788-849. noise_seq[0] is the initial draw, then one per step with t>0."""
    it = iter(noise_seq)
    x = next(it).clone()
    smoke_apply_conditions(x, padded_shape, init=init, control=control, **kw)
    for t_int in reversed(range(T)):
        x, _ = smoke_p_sample(model, buf, x, t_int, next(it) if t_int > 0 else None, guidance)
        smoke_apply_conditions(x, padded_shape, init=init, control=control, **kw)
    return x


def smoke_ddim_sample(model, buf, noise_seq, T, S, eta, *, padded_shape, init, control=None, guidance=None, **kw):
    """diffusion_2d.py:851-933."""
    it = iter(noise_seq)
    x = next(it).clone()
    smoke_apply_conditions(x, padded_shape, init=init, control=control, **kw)
    for time, time_next in ddim_times(T, S):
        t = torch.full((x.shape[0],), time, dtype=torch.long)
        eps, xs = smoke_model_predictions(model, buf, x, t, clip_x_start=True, rederive=True, guidance=guidance)
        if time_next < 0:
            x = xs
            continue
        x = ddim_update(buf, xs, eps, time, time_next, eta, next(it))
        smoke_apply_conditions(x, padded_shape, init=init, control=control, **kw)
    return x


# ----------------------------------------------------------------------------- Burgers (x is [B, C, H, W]), wavelet mode
def burgers_set_condition(img, u, shape, kind):
    """diffusion_1d.py:276-288 (is_wavelet branch)."""
    if kind == 'u0':
        img[:, -1, :u.shape[-2], :shape[-1]] = u[:, :, :shape[-1]]
    elif kind == 'uT':
        img[:, -1, -u.shape[-2]:, :shape[-1]] = u[:, :, :shape[-1]]
    elif kind == 'f':
        img[:, 4:8, :shape[-2], :shape[-1]] = u[:, :, :shape[-2], :shape[-1]]
    elif kind == 'low':
        img[:, 8:16, :shape[-2], :shape[-1]] = u[:, :, :shape[-2], :shape[-1]]
    elif kind == 'pad':
        img[:, :-1, shape[-2]:] = 0
        img[:, :, :, shape[-1]:] = 0
    else:
        raise ValueError(kind)


def burgers_apply_conditions(img, shape, flags, u0=None, uT=None, f=None, low=None):
    """Order used everywhere in diffusion_1d.py: pad, u0, uT, f, low."""
    if flags.get('pad'):
        burgers_set_condition(img, 0, shape, 'pad')
    if flags.get('u0'):
        burgers_set_condition(img, u0, shape, 'u0')
    if flags.get('uT'):
        burgers_set_condition(img, uT, shape, 'uT')
    if flags.get('f'):
        burgers_set_condition(img, f, shape, 'f')
    if flags.get('low'):
        burgers_set_condition(img, low, shape, 'low')


def burgers_p_losses(model, buf, x0, t, noise, *, padded_shape, loss_layer_weight, flags):
    """diffusion_1d.py:529-645 (pred_noise, base model). The target aliases `noise`, masked after x is built."""
    noise = noise.clone()
    nt = x0.shape[2]
    x = q_sample(buf, x0, t, noise)
    burgers_apply_conditions(x, padded_shape, flags, u0=x0[:, -1, :nt // 2, :], uT=x0[:, -1, nt // 2:, :], f=x0[:, 4:8],
                             low=x0[:, 8:16] if flags.get('low') else None)
    out = model(x, t)
    z = torch.zeros_like
    burgers_apply_conditions(noise, padded_shape, flags, u0=z(x[:, -1, :nt // 2, :]), uT=z(x[:, -1, nt // 2:, :]), f=z(x[:, 4:8]),
                             low=z(x0[:, 8:16]) if flags.get('low') else None)
    loss = (out - noise) ** 2 * loss_layer_weight
    loss = loss.reshape(loss.shape[0], -1).mean(dim=1) * buf['loss_weight'].gather(-1, t)
    return loss.mean()


def burgers_model_predictions(model, buf, x, t, clip_x_start=False, rederive=False):
    """diffusion_1d.py:205-227 without guidance (nablaJ == 0)."""
    eps = model(x, t)
    xs = predict_start_from_noise(buf, x, t, eps)
    if clip_x_start:
        xs = xs.clamp(-1., 1.)
        if rederive:
            eps = predict_noise_from_start(buf, x, t, xs)
    return eps, xs


def burgers_p_sample_loop(model, buf, noise_seq, T, *, padded_shape, flags, **cond):
    """diffusion_1d.py:310-373: conditions imposed before every model call and once after the loop."""
    it = iter(noise_seq)
    x = next(it).clone()
    for t_int in reversed(range(T)):
        burgers_apply_conditions(x, padded_shape, flags, **cond)
        t = torch.full((x.shape[0],), t_int, dtype=torch.long)
        _, xs = burgers_model_predictions(model, buf, x, t)
        xs = xs.clamp(-1., 1.)
        x = posterior_step(buf, x, t_int, xs, next(it) if t_int > 0 else None)
    burgers_apply_conditions(x, padded_shape, flags, **cond)
    return x


def burgers_ddim_sample(model, buf, noise_seq, T, S, eta, *, padded_shape, flags, **cond):
    """diffusion_1d.py:376-460."""
    it = iter(noise_seq)
    x = next(it).clone()
    for time, time_next in ddim_times(T, S):
        burgers_apply_conditions(x, padded_shape, flags, **cond)
        t = torch.full((x.shape[0],), time, dtype=torch.long)
        eps, xs = burgers_model_predictions(model, buf, x, t, clip_x_start=True, rederive=True)
        if time_next < 0:
            x = xs
            continue
        x = ddim_update(buf, xs, eps, time, time_next, eta, next(it))
    burgers_apply_conditions(x, padded_shape, flags, **cond)
    return x


# --------------------------------------------------------------------------- EMA of the weights (T1/T2 rows)
def ema_reference_update(ema, online, state, beta=0.995, update_every=10, update_after_step=100, inv_gamma=1.0, power=2.0 / 3.0,
                         min_value=0.0):
    """One `ema.update()` call of ema_pytorch.EMA as configured by both Trainers (train_diffusion.py:123-125,
    diffusion_2d.py:1165-1167). ema_pytorch is a third-party dependency that is absent from /root/reference (env.sh,
    unpinned) -> its published rule is restated; PARITY UNPINNED for this function (no reference output available).
    `ema`, `online`: dicts name -> tensor (ema is modified in place); `state`: {'step': int, 'initted': bool}."""
    step = state['step']
    state['step'] += 1
    if step % update_every != 0:
        return
    if step <= update_after_step:
        for k in ema:
            ema[k].copy_(online[k])
        return
    if not state['initted']:
        for k in ema:
            ema[k].copy_(online[k])
        state['initted'] = True
    epoch = max(state['step'] - update_after_step - 1, 0)
    decay = 0.0 if epoch <= 0 else min(max(1 - (1 + epoch / inv_gamma) ** -power, min_value), beta)
    for k in ema:
        ema[k].lerp_(online[k], 1 - decay)
