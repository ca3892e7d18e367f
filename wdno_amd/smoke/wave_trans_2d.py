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


def transform_simulation(X, smoke_out, wave_type='bior1.3', mode='zero', N_downsample=3):
    """The per-simulation body of the offline transform (wave_trans_2d.py:99-185) on the GPU.
    X [5, T, H, W]: density, two velocity and two control fields on the device; smoke_out [T]: fraction of smoke leaving
    through the target outlet. Returns the two dictionaries the script stores per simulation,
        (time_downsample file, space_downsample file) = {'coef', 'init_coef', 'smokeout', 'shape', 'ori_shape'}
    with level i holding the transform of the fields sub-sampled by 2^i in time / in space (tensors on the host, like the
    files). One 3-D DWT launch triple, one 2-D DWT and one 1-D DWT per level and variant."""
    from wdno_amd import wavelets as W
    xfm2d = W.DWTForward(J=1, mode=mode, wave=wave_type)
    xfm1d = W.DWT1DForward(J=1, mode=mode, wave=wave_type)
    s = smoke_out.reshape(1, 1, -1)
    files = []
    for kind in ('time', 'space'):
        coef, init, so = [], [], []
        for i in range(N_downsample):
            xs = (X[:, ::2 ** i] if kind == 'time' else X[:, :, ::2 ** i, ::2 ** i]).contiguous()
            coef.append(W.wavedec3_packed(xs, wave_type, mode).cpu())                     # [5, 8, T', H', W']
            yl0, yh0 = xfm2d(xs[:, :1].contiguous())
            init.append(torch.cat((yl0, yh0[0][:, 0]), dim=1).cpu())                       # [5, 4, H', W']
            ss = (s[:, :, ::2 ** i] if kind == 'time' else s).contiguous()
            lo, hi = xfm1d(ss)
            so.append(torch.cat((lo, hi[0]), dim=1).cpu()[0])                              # [2, T']
        files.append({'coef': coef, 'init_coef': init, 'smokeout': so, 'shape': [c.shape[-3:] for c in coef], 'ori_shape': X.shape[1:]})
    return files[0], files[1]
