"""Helpers with the names burgers/ddpm_burgers/model_utils.py exports (extract, schedules, small utilities, the
guidance-gradient wrapper). Host-side logic only."""
import math

import torch

from wdno_amd.diffusion_core import cosine_beta_schedule, linear_beta_schedule  # noqa: F401  (re-exported)


def normalize_to_neg_one_to_one(img):
    return img * 2 - 1


def unnormalize_to_zero_to_one(t):
    return (t + 1) * 0.5


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def identity(t, *args, **kwargs):
    return t


def cycle(dl):
    while True:
        for data in dl:
            yield data


def extract(a, t, x_shape):
    """a[t] broadcast to [B, 1, ...] (model_utils.py:27-30)."""
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


def get_nablaJ(loss_fn):
    """Gradient of a scalar-per-sample guidance loss w.r.t. the sample (model_utils.py:35-50)."""
    def nablaJ(x):
        x.requires_grad_(True)
        J = loss_fn(x)
        return torch.autograd.grad(J, x, grad_outputs=torch.ones_like(J), retain_graph=True, create_graph=True, allow_unused=True)[0].detach()
    return nablaJ
