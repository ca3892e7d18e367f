"""Datasets of the 1-D Burgers task -- drop-in for burgers/ddpm_burgers/data_burgers_1d.py.

`get_wavelet_super_preprocess` turns the coefficient file of the offline transform
    {'coef': [level][N, 2, 4, h, w], 'shape', 'ori_shape'}
into U-Net inputs [N, C, 64, 64]: channels 0-3 the sub-bands of u, 4-7 those of f, zero-padded from 41 x 60 (8 more channels
with the nearest-upsampled next-coarser level for super-resolution models), and one condition channel that holds the 1-D DWT
of u(t=0) and u(t=T) in four horizontal stripes; everything divided by the per-channel rescaler. The condition channel needs
the physical u: IDWT of the packed coefficients, then a 1-D DWT of two rows -- both run on the GPU here (the reference used
pytorch_wavelets on whatever device the file was loaded to); the rest is indexing. Results come back on the device of the
input file, so `DiffusionDataset` / `DataLoader` behave as before.
"""
import math
import random

import torch
from torch import nn
from torch.utils.data import DataLoader, Dataset

from ddpm_burgers.model_utils import cycle
from ddpm_burgers.wavelet_utils import upsample_coef  # noqa: F401  (re-export)
from wave_trans import tensor_to_coef

_MAX_SAMPLES = 40000


def _nearest2(t):
    """nearest x2 in (t, x) of [N, l, nt, nx] (ddpm_burgers.wavelet_utils.upsample_coef): indexing only, any device"""
    return t.repeat_interleave(2, dim=-2).repeat_interleave(2, dim=-1)


def _gpu_chunks(n, chunk=2048):
    return [(a, min(n, a + chunk)) for a in range(0, n, chunk)]


def _condition_rows(w_uf, shape, ori_shape, mode, wave_type):
    """1-D DWT of u(t=0) and u(t=T) from packed coefficients [N, >=8, 64, 64]: -> (lo [N, 2, nx], hi [N, 2, nx]) on w_uf's device"""
    from wdno_amd import wavelets as W
    ifm = W.DWTInverse(mode=mode, wave=wave_type)
    xfm1d = W.DWT1DForward(J=1, mode=mode, wave=wave_type)
    dev = w_uf.device
    gpu = torch.device('cuda', torch.cuda.current_device())
    los, his = [], []
    for a, b in _gpu_chunks(w_uf.shape[0]):
        yl, yh = tensor_to_coef(w_uf[a:b, :8].to(gpu), shape)
        u = ifm((yl.contiguous(), [yh[0].contiguous()]))[:, 0, :ori_shape[-2], :ori_shape[-1]]
        lo, hi = xfm1d(u[:, [0, -1], :ori_shape[-1]].contiguous())
        los.append(lo.to(dev))
        his.append(hi[0].to(dev))
    return torch.cat(los), torch.cat(his)


def get_wavelet_super_preprocess(rescaler=70, is_super_model=False, N_downsample=0, mode='zero', wave_type='bior2.4',
                                 is_condition_u0=True, is_condition_uT=True):
    if rescaler is None:
        raise NotImplementedError('Should specify rescaler. If no rescaler is not used, specify 1.')

    def preprocess(db):
        lvl = N_downsample if is_super_model else 0
        data = db['coef']
        w_u, w_f = data[lvl][:, 0][:_MAX_SAMPLES], data[lvl][:, 1][:_MAX_SAMPLES]
        ori_shape = list(db['ori_shape'])
        ori_shape[0] = math.ceil(ori_shape[0] / 2 ** lvl)
        ori_shape[1] = math.ceil(ori_shape[1] / 2 ** lvl)
        n, nt, nx = w_u.size(0), w_f.size(-2), w_f.size(-1)
        shape = w_f.shape[2:]
        pad_t = pad_x = int(64 / 2 ** lvl)
        w_uf = nn.functional.pad(torch.cat((w_u, w_f), dim=1), (0, pad_x - nx, 0, pad_t - nt))          # [N, 8, pad, pad]
        out = w_uf
        if is_super_model:
            sub = torch.cat((_nearest2(data[lvl + 1][:, 0][:_MAX_SAMPLES]), _nearest2(data[lvl + 1][:, 1][:_MAX_SAMPLES])), dim=1)
            sub = nn.functional.pad(sub, (0, pad_x - sub.shape[-1], 0, pad_t - sub.shape[-2]))
            w_uf[:, :, nt, :] = w_uf[:, :, nt - 1, :]            # odd number of time coefficients: repeat the last one
            out = torch.cat((w_uf, sub), dim=1)
        if is_condition_u0 or is_condition_uT:
            lo, hi = _condition_rows(w_uf, shape, ori_shape, mode, wave_type)
            cond = torch.zeros_like(out[:, :1])
            rep = pad_t // 4
            if is_condition_u0:
                cond[:, 0, :rep, :nx] = lo[:, None, 0]
                cond[:, 0, rep:2 * rep, :nx] = hi[:, None, 0]
            if is_condition_uT:
                cond[:, 0, 2 * rep:3 * rep, :nx] = lo[:, None, 1]
                cond[:, 0, 3 * rep:4 * rep, :nx] = hi[:, None, 1]
            out = torch.cat((out, cond), dim=1)
        return out / rescaler, list(shape), list(ori_shape)

    return preprocess


def get_burgers_preprocess(rescaler=10, is_super_model_train=False, N_downsample=0, is_super_model_test=False, upsample_t=0, upsample_x=0):
    """Physical-space (non-wavelet) packing, data_burgers_1d.py:168-211: u, f zero-padded to 128 x 128 (times 2^upsample)."""
    if rescaler is None:
        raise NotImplementedError('Should specify rescaler. If no rescaler is not used, specify 1.')

    def preprocess(db):
        if is_super_model_test:
            super_nt, super_nx = db['f'].shape[-2], db['u'].shape[-1]
            assert super_nt / 80 / 2 ** upsample_t > 0
            st, sx = int(super_nt / 80 / 2 ** upsample_t), int(super_nx / 120 / 2 ** upsample_x)
            u, f = db['u'][:, ::st, ::sx], db['f'][:, ::st, ::sx]
        else:
            u, f = db['u'][:_MAX_SAMPLES], db['f'][:_MAX_SAMPLES]
        nt, nx = f.size(-2), f.size(-1)
        shape = u[..., ::2 ** N_downsample, ::2 ** N_downsample].shape[-2:]
        f = nn.functional.pad(f, (0, 128 * 2 ** upsample_x - nx, 0, 128 * 2 ** upsample_t - nt))
        u = nn.functional.pad(u, (0, 128 * 2 ** upsample_x - nx, 0, 128 * 2 ** upsample_t - 1 - nt))
        data = torch.stack((u, f), dim=1)
        if is_super_model_train:
            uf = data[:, :, ::2 ** N_downsample, ::2 ** N_downsample]
            uf_sub = _nearest2(data[:, :, ::2 ** (N_downsample + 1), ::2 ** (N_downsample + 1)])
            nt_sub = int(nt / 2 ** N_downsample)
            uf[:, :, nt_sub + 1, :] = uf[:, :, nt_sub, :]
            data = torch.cat((uf, uf_sub), dim=1)
        return data / rescaler, list(shape), list(shape)

    return preprocess


def get_wavelet_preprocess(rescaler=70, mode='zero', wave_type='bior2.4', is_condition_u0=True, is_condition_uT=True):
    """Multi-level (J = len(shape)) variant of the packer, data_burgers_1d.py:90-166. train_ddpm_burgers.py never selects it
    (it only appears as the default argument of DiffusionDataset); the single-level packer above is the WDNO path."""
    if rescaler is None:
        raise NotImplementedError('Should specify rescaler. If no rescaler is not used, specify 1.')

    def preprocess(db):
        raise NotImplementedError('multi-level coefficient datasets are not produced by wave_trans.py; use get_wavelet_super_preprocess')

    return preprocess


class DiffusionDataset(Dataset):
    def __init__(self, fname, preprocess=None):
        self.db = torch.load(fname, weights_only=False) if isinstance(fname, str) else fname
        self.x, self.shape, self.ori_shape = (preprocess or get_wavelet_preprocess())(self.db)

    def __len__(self):
        return self.x.size(0)

    def __getitem__(self, idx):
        return self.x[idx]

    def get(self, idx):
        return self.__getitem__(idx)

    def len(self):
        return self.__len__()


class SuperDataLoader:
    """One batch per iteration from a randomly chosen dataset of the list (super-resolution levels)."""

    def __init__(self, dataset, batch_size=1, shuffle=True, pin_memory=True, num_workers=1, seed=None):
        """seed (an extension): private shuffle / level-choice streams, e.g. base + 1000 * rank under data parallelism -- the
        reference's loader is not sharded by accelerate, and identically seeded ranks would draw identical batches."""
        self.dataset = dataset
        gens = [None if seed is None else torch.Generator().manual_seed(seed + i) for i in range(len(dataset))]
        self.dl = [cycle(DataLoader(ds, batch_size=batch_size, shuffle=shuffle, pin_memory=pin_memory, num_workers=num_workers, generator=g))
                   for ds, g in zip(dataset, gens)]
        self.num_batches = len(dataset) * ((len(dataset[0]) + batch_size - 1) // batch_size)
        self._rng = random if seed is None else random.Random(seed)

    def __iter__(self):
        yield next(self.dl[self._rng.randint(0, len(self.dl) - 1)])

    def __len__(self):
        return self.num_batches
