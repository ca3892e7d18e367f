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


def _split_row(rows):
    """inference_2d.py:44-45 writes int(40/2) whatever the tensor size: the super-resolution tensors (80 x 80) split at row 20 too. Tensors
    of 20 rows or fewer (the reduced-size tests; the reference's second mean would be over nothing there: NaN) are halved instead."""
    return 20 if rows > 20 else rows // 2


def guidance_value(x, shape, ori_shape, rescaler, wave_type='bior1.3', pad_mode='zero', is_condition_control=False, w_energy=0.0, w_init=0.0,
                   init_u=None):
    """The scalar J above (differentiable w.r.t. x). x [B, F, 42, 40, 40] in rescaled (network) units; rescaler [1, 1, 42, 1, 1]."""
    xs = x * rescaler
    coef = tensor_to_coef(xs[:, :, :-2].permute(0, 2, 1, 3, 4), shape)
    rec = W.waverec3([coef[0].contiguous(), {k: v.contiguous() for k, v in coef[1].items()}], wave_type, pad_mode)
    state = rec[:, :ori_shape[0], :ori_shape[1], :ori_shape[2]].reshape(-1, 5, ori_shape[0], ori_shape[1], ori_shape[2])
    half = _split_row(xs.shape[-2])
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


# ----------------------------------------------------------------------------------------------------- closed form (no autograd)
_success_cache = {}


def _success_gradient(t_coef, t_ori, wave_type, pad_mode, device):
    """d smoke_out[T-1] / d(lo, hi): two constant vectors of length T' (the 1-D synthesis is linear), computed once."""
    key = (t_coef, t_ori, wave_type, pad_mode, str(device))
    if key not in _success_cache:
        with torch.enable_grad():
            lo = torch.zeros(1, 1, t_coef, device=device, requires_grad=True)
            hi = torch.zeros(1, 1, t_coef, device=device, requires_grad=True)
            so = W.DWT1DInverse(mode=pad_mode, wave=wave_type)((lo, [hi]))[:, 0]
            g_lo, g_hi = torch.autograd.grad(so[:, t_ori - 1].sum(), (lo, hi))
        _success_cache[key] = (g_lo.reshape(-1).contiguous(), g_hi.reshape(-1).contiguous())
    return _success_cache[key]


def guidance_fn_explicit(x, shape, ori_shape, rescaler, wave_type='bior1.3', pad_mode='zero', is_condition_control=False, w_energy=0.0,
                         w_init=0.0, init_u=None):
    """The same gradient as guidance_fn, written out: J is quadratic in the reconstructed state and linear in the smoke-out
    channel, so  dJ/dxs = unpack^T( IDWT3^T( dJ/dstate ) ) + (constant for the smoke-out channel)  with
        dJ/dstate[:, 0, 0]  = 2 w_init (state[:, 0, 0] - init_u) / (H W),      dJ/dstate[:, 3:5] = 2 w_energy state[:, 3:5] / (2 T H W).
    One synthesis launch, one adjoint-synthesis launch (wdno_dwt_inv_adjoint) and index work; no autograd tape, no host sync --
    the guided sampling step can therefore be captured in a HIP graph like the unguided one (diffusion_core._StepGraph)."""
    from wdno_amd import ops
    from wdno_amd.wavelets import MODES, _filters
    xs = (x * rescaler).contiguous()
    b, f, c, hh, ww = xs.shape
    tc, hc, wc = (int(v) for v in shape)
    to, ho, wo = (int(v) for v in ori_shape)
    # unpack: [B, F, 40, H, W] -> packed coefficients [B*5, 8, T', H', W'] (tensor_to_coef + coef_to_tensor order)
    packed = xs[:, :tc, :40, :hc, :wc].reshape(b, tc, 5, 8, hc, wc).permute(0, 2, 3, 1, 4, 5).reshape(b * 5, 8, tc, hc, wc).contiguous()
    filt, L = _filters(wave_type)
    m = MODES[pad_mode]
    sig = [2 * v if m == 0 else 2 * v - L + 2 for v in (tc, hc, wc)]
    cs = (8 * tc * hc * wc, tc * hc * wc, hc * wc, wc)
    rec = torch.empty((b * 5, *sig), device=xs.device, dtype=torch.float32)
    ops.dwt_call('inv', packed, rec, 3, m, filt, b * 5, sig, [tc, hc, wc], cs)
    state = rec.reshape(b, 5, *sig)[:, :, :to, :ho, :wo]
    drec = torch.zeros_like(rec).reshape(b, 5, *sig)
    if init_u is not None and w_init != 0.0:
        drec[:, 0, 0, :ho, :wo] = (state[:, 0, 0] - init_u.to(state.device)) * (2.0 * w_init / (ho * wo))
    if not is_condition_control and w_energy != 0.0:
        drec[:, 3:5, :to, :ho, :wo] = state[:, 3:5] * (2.0 * w_energy / (2 * to * ho * wo))
    dpacked = torch.empty_like(packed)
    ops.dwt_call('inv_adjoint', drec.reshape(b * 5, *sig), dpacked, 3, m, filt, b * 5, sig, [tc, hc, wc], cs)
    g = torch.zeros_like(xs)
    g[:, :tc, :40, :hc, :wc] = dpacked.reshape(b, 5, 8, tc, hc, wc).permute(0, 3, 1, 2, 4, 5).reshape(b, tc, 40, hc, wc)
    if not is_condition_control:
        g_lo, g_hi = _success_gradient(tc, to, wave_type, pad_mode, xs.device)
        half = _split_row(hh)                  # the reference halves the ROW axis of the smoke-out channel (x[:, :T', -1, :20])
        g[:, :tc, -1, :half, :] -= (g_lo / (half * ww)).reshape(1, tc, 1, 1)
        g[:, :tc, -1, half:, :] -= (g_hi / ((hh - half) * ww)).reshape(1, tc, 1, 1)
    return g


class GuidanceFn:
    """`design_fn` for GaussianDiffusion.sample (called as design_fn(x, low=, init=, init_u=), inference_2d.py:30-66) on the explicit
    gradient. `graph_safe = True` tells the sampler that the callback launches only capturable work."""
    graph_safe = True

    def __init__(self, shape, ori_shape, rescaler, **kw):
        self.shape, self.ori_shape, self.rescaler, self.kw = shape, ori_shape, rescaler, kw

    def __call__(self, x, low=None, init=None, init_u=None):
        return guidance_fn_explicit(x, self.shape, self.ori_shape, self.rescaler, init_u=init_u, **self.kw)
