"""Coefficient <-> tensor packing helpers of burgers/wave_trans.py:18-62.

With the HIP transform the packed tensor is produced directly (wdno_amd.wavelets.DWTForward.packed); these helpers
keep the reference's names and argument order for callers that hold (Yl, Yh) pairs. They are views / concatenations
only -- no arithmetic."""
import torch
from torch import nn

from ddpm_burgers.wavelet_utils import upsample_coef  # noqa: F401  (re-export, as in the reference module)


def tensor_to_coef(coef_tensor, shape):
    """[N, >= 8, Hp, Wp] -> (Yl [N, 2, h, w], [Yh [N, 2, 3, h, w]]) cropping the zero padding."""
    h, w = int(shape[-2]), int(shape[-1])
    u = coef_tensor[:, 0:4, :h, :w]
    f = coef_tensor[:, 4:8, :h, :w]
    yl = torch.stack((u[:, 0], f[:, 0]), dim=1)
    yh = [torch.stack((u[:, 1:4], f[:, 1:4]), dim=1)]
    return yl, yh


tensor_to_coef_super = tensor_to_coef


def coef_to_tensor(Yl, Yh, pad=False):
    """(Yl [N,C,h,w], [Yh [N,C,3,h,w]]) -> [N, C, 1+3J, h, w]; pad=True zero-pads to multiples of 64 x 64 like the
    reference. For J > 1 every band is nearest-repeated to the finest grid (Yl by 2^(J-1), Yh[i] by 2^i) and the
    row deficit of the coarser bands is filled by replicating their last row (wave_trans.py:50-57)."""
    J = len(Yh)
    if J == 1:
        t = torch.cat((Yl.unsqueeze(2), Yh[0]), dim=2)
    else:
        top = 2 ** (J - 1)
        rows, cols = Yh[0].shape[-2] + top - 1, Yh[0].shape[-1]
        t = torch.zeros(Yl.shape[0], Yl.shape[1], 1 + 3 * J, rows, cols, device=Yl.device, dtype=Yl.dtype)
        t[:, :, 0] = Yl.repeat_interleave(top, dim=-2).repeat_interleave(top, dim=-1)
        for i in range(J):
            r = Yh[i].repeat_interleave(2 ** i, dim=-2).repeat_interleave(2 ** i, dim=-1)
            t[:, :, 1 + 3 * i:4 + 3 * i, :r.shape[-2]] = r
            if top - 2 ** i:
                t[:, :, 1 + 3 * i:4 + 3 * i, r.shape[-2]:] = r[:, :, :, -1:]
    if pad:
        up_t = int(t.shape[-2] / 40)
        up_x = int(t.shape[-1] / 60)
        t = nn.functional.pad(t, (0, 64 * up_x - t.shape[-1], 0, 64 * up_t - t.shape[-2]), 'constant', 0)
    return t


def transform_dataset(data, wave_type='bior2.4', mode='periodization', N_downsample=4):
    """The body of the offline transform (wave_trans.py:99-123) on the GPU: data [N, 2, nt, nx] = (u, f with its last row
    zero-filled) -> {'coef': [level][N, 2, 4, h, w], 'shape', 'ori_shape'} with level i the single-level 2-D DWT of the
    fields sub-sampled by 2^i in t and x (the file train_ddpm_burgers.py reads as coef_<wave>_<mode>_super)."""
    from wdno_amd import wavelets as W
    coef = [W.dwt_packed(data[:, :, ::2 ** i, ::2 ** i].contiguous(), wave_type, mode, 2).cpu() for i in range(N_downsample)]
    return {'coef': coef, 'shape': [c.shape[2:] for c in coef], 'ori_shape': data.shape[2:]}
