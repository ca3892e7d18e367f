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


# ----------------------------------------------------------------------------------------------------- guidance step-size schedules
# model_utils.py:52-138: eta(t) tables over the 1000 sampling steps, indexed by the integer timestep.
_SAMPLING_STEPS = 1000


def _betas_from_alphas_cumprod(ac):
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


def cosine_beta_J_schedule(t, s=0.008):
    """Increasing cosine-beta table (model_utils.py:90-101)."""
    x = torch.linspace(0, _SAMPLING_STEPS, _SAMPLING_STEPS + 1, dtype=torch.float64)
    return _betas_from_alphas_cumprod(torch.cos((x / _SAMPLING_STEPS + s) / (1 + s) * math.pi * 0.5) ** 2)[t]


def plain_cosine_schedule(t, s=0.0):
    """cos((x + s) / (T + s)) read backwards (model_utils.py:103-111; the reference calls Tensor.flip() without dims,
    which raises -- the evident intent, a flip along the only axis, is implemented)."""
    x = torch.linspace(0, _SAMPLING_STEPS, _SAMPLING_STEPS + 1, dtype=torch.float64)
    return torch.cos((x + s) / (_SAMPLING_STEPS + s)).flip(0)[t]


def sigmoid_schedule(t, start=-3, end=3, tau=1, clamp_min=1e-5):
    """Sigmoid-beta table (model_utils.py:113-127)."""
    x = torch.linspace(0, _SAMPLING_STEPS, _SAMPLING_STEPS + 1, dtype=torch.float64) / _SAMPLING_STEPS
    v_start, v_end = torch.tensor(start / tau).sigmoid(), torch.tensor(end / tau).sigmoid()
    return _betas_from_alphas_cumprod((v_end - ((x * (end - start) + start) / tau).sigmoid()) / (v_end - v_start))[t]


def sigmoid_schedule_flip(t):
    return sigmoid_schedule(999 - t)


def linear_schedule(t):
    """model_utils.py:132-137 (a one-entry table: beta_start * 1000)."""
    return torch.linspace(1000 * 0.0001, 1000 * 0.02, 1, dtype=torch.float64)[t]


def get_scheduler(scheduler):
    """Name -> schedule function (model_utils.py:52-68)."""
    if scheduler is None:
        return None
    if scheduler == 'linear':
        raise NotImplementedError
    table = {'cosine': cosine_beta_J_schedule, 'plain_cosine': plain_cosine_schedule, 'sigmoid': sigmoid_schedule,
             'sigmoid_flip': sigmoid_schedule_flip}
    if scheduler not in table:
        raise ValueError
    return table[scheduler]


def get_proj_ep_orthogonal_func(norm='F'):
    """Guidance combiner  eps + grad - <grad, eps> eps / ||eps||  (model_utils.py:70-87). The inner product runs over the whole
    batch of fields for 'F' (with the per-field Frobenius norm of eps), over x for '1D_x', over t for '1D_t'."""
    if norm not in ('F', '1D_x', '1D_t'):
        raise NotImplementedError

    def proj_ep_orthogonal(ep, nabla_J):
        prod, sq = nabla_J * ep, ep.square()
        if norm == 'F':
            inner, length = prod.sum(), sq.sum((-2, -1), keepdim=True).sqrt()
        elif norm == '1D_x':
            inner, length = prod.sum(-1, keepdim=True), sq.sum(-1, keepdim=True).sqrt()
        else:                              # '1D_t': reduced over t, broadcast back along x
            inner, length = prod.sum(-2), sq.sum(-2).sqrt()
        return ep + nabla_J - inner * ep / length
    return proj_ep_orthogonal


# ----------------------------------------------------------------------------------------------------- small helpers (model_utils.py:163-177)
def has_int_squareroot(num):
    return (math.sqrt(num) ** 2) == num


def num_to_groups(num, divisor):
    full, rest = divmod(num, divisor)
    return [divisor] * full + ([rest] if rest > 0 else [])


def convert_image_to_fn(img_type, image):
    return image.convert(img_type) if image.mode != img_type else image
