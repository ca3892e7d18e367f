"""ORACLE (test infrastructure only) -- the smoke control objective whose gradient guides the sampler.

Only tests/ may import this file; the product path (wdno_amd/smoke/guidance.py) never does.

Restates smoke/inference_2d.py:30-66 (wavelet parametrisation) in numpy fp64 on top of oracle/dwt_ref.py:
    x' = x * RESCALER
    state     = waverec3(tensor_to_coef(x'[:, :, :-2]))[:, :T, :H, :W]  as [B, 5, T, H, W]         (inference_2d.py:40-42)
    smoke_out = IDWT1( mean over rows [0, 20) of the last channel, mean over rows [20, end) )      (inference_2d.py:43-46: int(40/2))
    J = - sum_b smoke_out[b, T-1] + w_energy * sum_b mean(state[b, 3:5]^2) + w_init * sum_b mean((state[b, 0, 0] - init_u)^2)
    (with is_condition_control only the w_init term remains, inference_2d.py:51-54)
The reference returns dJ/dx' by autograd. J is at most quadratic in x', so a central difference of J along any direction v equals
<dJ/dx', v> exactly (up to fp64 round-off); `directional_derivative` is that difference.

Parity pin: tests/golden/ref_guidance.npz holds gradients RETURNED BY THE REFERENCE'S guidance_fn (generator
tests/golden/make_ref_guidance_golden.py); tests/test_oracle_dwt.py::test_guidance_objective_vs_reference_run checks this file against them.
"""
import numpy as np

from . import dwt_ref as R

SPLIT_ROW = 20


def objective(xs, shape, ori_shape, init_u, w_energy=0.0, w_init=0.0, is_condition_control=False, wave='bior1.3', mode='zero'):
    """J(x') for x' = x * RESCALER [B, F, 42, H, W] (numpy fp64)."""
    tc = shape[0]
    to, ho, wo = ori_shape
    lll, det = R.smoke_tensor_to_coef(np.transpose(xs[:, :, :-2], (0, 2, 1, 3, 4)), shape)
    state = R.idwt3(lll, det, wave, mode)[:, :to, :ho, :wo].reshape(-1, 5, to, ho, wo)
    g_init = ((state[:, 0, 0] - init_u) ** 2).mean((-1, -2)).sum()
    if is_condition_control:
        return w_init * g_init
    lo = xs[:, :tc, -1, :SPLIT_ROW].mean((-2, -1))[:, None]
    hi = xs[:, :tc, -1, SPLIT_ROW:].mean((-2, -1))[:, None]
    smoke_out = R.idwt1d(lo, hi, wave, mode)[:, 0]
    return -smoke_out[:, to - 1].sum() + w_energy * (state[:, 3:5] ** 2).mean((1, 2, 3, 4)).sum() + w_init * g_init


def directional_derivative(xs, v, *a, **k):
    """<dJ/dx', v>, exact for the (at most quadratic) objective."""
    return objective(xs + 0.5 * v, *a, **k) - objective(xs - 0.5 * v, *a, **k)
