"""Guidance gradient of the smoke control task on the GPU (smoke/inference_2d.py:30-66, wavelet parametrisation).

The sampler calls `design_fn(x)` once per step (diffusion_2d.py:733-741); the reference evaluates
    x' = x * RESCALER ; state = IDWT3(unpack(x')) ; smoke_out = IDWT1(mean of the two halves of the last channel)
    J = - sum_b smoke_out[b, T-1] + w_energy * sum_b mean(state[b, 3:5]^2) + w_init * sum_b mean((state[b, 0, 0] - init_u)^2)
(the first two terms drop out when the model is conditioned on the control) and returns dJ/dx'. Here the same expression is
written on the HIP transforms, whose backward passes are the exact adjoint kernels (wdno_dwt_inv_adjoint), so the gradient
costs one synthesis and one adjoint-synthesis launch triple per call.
"""
import torch

from wdno_amd import wavelets as W
from wave_trans_2d import tensor_to_coef


def guidance_value(x, shape, ori_shape, rescaler, wave_type='bior1.3', pad_mode='zero', is_condition_control=False, w_energy=0.0, w_init=0.0,
                   init_u=None):
    """The scalar J above (differentiable w.r.t. x). x [B, F, 42, 40, 40] in rescaled (network) units; rescaler [1, 1, 42, 1, 1]."""
    xs = x * rescaler
    coef = tensor_to_coef(xs[:, :, :-2].permute(0, 2, 1, 3, 4), shape)
    rec = W.waverec3([coef[0].contiguous(), {k: v.contiguous() for k, v in coef[1].items()}], wave_type, pad_mode)
    state = rec[:, :ori_shape[0], :ori_shape[1], :ori_shape[2]].reshape(-1, 5, ori_shape[0], ori_shape[1], ori_shape[2])
    half = xs.shape[-1] // 2
    lo = xs[:, :shape[0], -1, :half].mean((-2, -1)).unsqueeze(1)
    hi = xs[:, :shape[0], -1, half:].mean((-2, -1)).unsqueeze(1)
    smoke_out = W.DWT1DInverse(mode=pad_mode, wave=wave_type)((lo.contiguous(), [hi.contiguous()]))[:, 0]
    g_init = (state[:, 0, 0] - init_u.to(state.device)).square().mean((-1, -2)).sum() if init_u is not None else 0.0
    if is_condition_control:
        return w_init * g_init
    g_success = smoke_out[:, ori_shape[0] - 1].sum()
    g_energy = state[:, 3:5].square().mean((1, 2, 3, 4)).sum()
    return -g_success + w_energy * g_energy + w_init * g_init


def guidance_fn(x, shape, ori_shape, rescaler, **kw):
    """dJ/d(x * RESCALER), the quantity inference_2d.guidance_fn returns (the sampler scales it, diffusion_2d.py:733-741)."""
    with torch.enable_grad():
        xs = (x.detach() * rescaler).requires_grad_(True)
        j = guidance_value(xs, shape, ori_shape, torch.ones_like(rescaler), **kw)
        return torch.autograd.grad(j, xs)[0]
