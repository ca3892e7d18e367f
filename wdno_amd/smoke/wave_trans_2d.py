"""Coefficient <-> tensor packing helpers of smoke/wave_trans_2d.py:17-58 (views / concatenations only)."""
import torch

BANDS = ('aad', 'ada', 'add', 'daa', 'dad', 'dda', 'ddd')


def tensor_to_coef(coef_tensor, shape, upsample_type=None):
    """[N, >= 40, Tp, Hp, Wp] -> (Yl [N*5, t, h, w], {band: [N*5, t, h, w]}); 'time' / 'space' skip one leading
    coefficient on the up-sampled axes (super-resolution models)."""
    t, h, w = int(shape[-3]), int(shape[-2]), int(shape[-1])
    t0 = 1 if upsample_type == 'time' else 0
    s0 = 1 if upsample_type == 'space' else 0
    c = coef_tensor[:, :40, t0:t0 + t, s0:s0 + h, s0:s0 + w]
    c = c.reshape(c.shape[0], 5, 8, t, h, w)
    yl = c[:, :, 0].reshape(-1, t, h, w)
    yh_ = c[:, :, 1:].reshape(-1, 7, t, h, w)
    return yl, {k: yh_[:, i] for i, k in enumerate(BANDS)}


def coef_to_tensor(coef, pad=False):
    yl, yh = coef[0], coef[1]
    return torch.cat((yl[:, None], torch.stack(list(yh.values()), dim=1)), dim=1)
