"""Wavelet-coefficient dataset of the 2-D smoke task -- drop-in for smoke/ddpm/data_2d.py:119-232 (Smoke_wave, SuperDataLoader).

A simulation is stored by the offline transform (wave_trans_2d.py, here: wave_trans_2d.transform_simulation) as
    {'coef': [level][5, 8, T', H', W'], 'init_coef': [level][5, 4, H', W'], 'smokeout': [level][2, T'], 'shape', 'ori_shape'}
and becomes one U-Net input [frames, channels, 40, 40]:
    channels 0..39   the 5 fields x 8 sub-bands, zero-padded to 24 x 40 x 40
    (super-resolution models: 40 more channels with the next-coarser level, nearest-upsampled in time or in space, the fine
     level carrying one replicated border coefficient so that both grids line up)
    channel -2       initial-density condition: the 4 sub-bands of the 2-D DWT of rho(t=0), each repeated over frames/4 frames
    channel -1       smoke-out condition: the 2 sub-bands of its 1-D DWT painted over the upper / lower half of the image
all divided by the per-channel RESCALER. `pack_smoke_state` is the packing itself and works on tensors of any device, so a
batch can be packed on the GPU straight from transform_simulation's output; `Smoke_wave` keeps the reference's file-per-item
interface. The raw-field dataset `Smoke` is not on the WDNO path: it resolves to the reference module when that is present.
"""
import math
import os
import random

import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset

import wdno_amd

_RESCALERS = {
    'bior2.2': [4, 2, 2, 1, 2, 2, 1, 1, 42, 10, 21, 8, 15, 3, 5, 2, 51, 18, 8, 5, 16, 6, 4, 2,
                42, 8, 17, 6, 15, 3, 5, 2, 51, 18, 9, 5, 13, 5, 3, 2, 3, 2],
    'bior1.3': [4, 2, 2, 2, 2, 2, 1, 1, 37, 12, 15, 11, 19, 6, 11, 5, 44, 24, 9, 10, 16, 9, 6, 6,
                37, 10, 15, 8, 19, 5, 11, 5, 43, 24, 9, 10, 16, 9, 5, 5, 3, 2],
}


def cycle(dl):
    while True:
        for data in dl:
            yield data


def pack_smoke_state(coef, init_coef, smokeout, rescaler, coef_sub=None, downsample_type='time', pad_t=24, pad_x=40):
    """coef [5, 8, t, x, x] (+ coef_sub: the next-coarser level for super-resolution models), init_coef [5, 4, x, x],
    smokeout [2, t]  ->  state [pad_t, C, pad_x, pad_x] / rescaler, C = 42 (or 82)."""
    nt, nx = coef.shape[-3], coef.shape[-1]
    w = coef.reshape(40, nt, nx, nx)
    data = F.pad(w, (0, pad_x - nx, 0, pad_x - nx, 0, pad_t - nt))
    if coef_sub is not None:
        w_sub = coef_sub.reshape(40, *coef_sub.shape[-3:])
        kind = 'space' if downsample_type == 'space' else 'time'
        # nearest x2 of the coarse level (ddpm/wave_utils.upsample_coef): pure indexing, valid on host and device tensors
        w_sub = w_sub.repeat_interleave(2, dim=1) if kind == 'time' else w_sub.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        if kind == 'space':
            fine = F.pad(w, (1, 1, 1, 1), mode='replicate')
        else:
            fine = torch.cat((w[:, :1], w, w[:, -1:]), dim=1)
        assert fine.shape == w_sub.shape, (fine.shape, w_sub.shape)
        both = torch.cat((fine, w_sub), dim=0)
        data = F.pad(both, (0, pad_x - both.shape[-1], 0, pad_x - both.shape[-2], 0, pad_t - both.shape[-3]))
    # initial density: its 4 sub-bands, each shown for pad_t / 4 consecutive frames
    rho0 = init_coef[0]
    cond = rho0.unsqueeze(1).expand(rho0.shape[0], pad_t // 4, nx, nx).reshape(-1, nx, nx)
    cond = F.pad(cond, (0, pad_x - nx, 0, pad_x - nx))
    # smoke-out: approximation band over the upper half of the image, detail band over the lower half
    assert pad_x % 2 == 0
    so = smokeout.permute(1, 0)                                            # [t, 2]
    so = so.reshape(1, so.shape[0], 2, 1, 1).expand(1, so.shape[0], 2, pad_x // 2, pad_x).reshape(1, so.shape[0], pad_x, pad_x)
    so = F.pad(so, (0, 0, 0, 0, 0, pad_t - nt))
    state = torch.cat((data, cond.unsqueeze(0), so), dim=0)
    return state.permute(1, 0, 2, 3) / rescaler


def pack_smoke_batch(coef, init_coef, smokeout, rescaler, pad_t=24, pad_x=40):
    """pack_smoke_state for a whole batch of base-resolution simulations held on one device (the GPU-side packer of SURVEY 8f-1: fields
    -> HIP DWT -> this -> train step, no per-sample torch.load / host packing): coef [B, F, 8, t, x, x] (F fields), init_coef
    [B, 4, x, x] (the sub-bands of the 2-D DWT of rho(t=0)), smokeout [B, 2, t]  ->  states [B, pad_t, 8 F + 2, pad_x, pad_x] /
    rescaler. Index work only; equal to stacking pack_smoke_state per sample (tests/test_host.py)."""
    b, nf = coef.shape[0], coef.shape[1]
    nt, nx = coef.shape[-3], coef.shape[-1]
    if coef.is_cuda:                     # one launch of csrc/pack.hip (bit-identical to the index work below, tests/test_gpu_data.py)
        return pack_smoke_gpu(coef, init_coef, smokeout, rescaler, None, pad_t, pad_x)
    data = F.pad(coef.reshape(b, nf * 8, nt, nx, nx), (0, pad_x - nx, 0, pad_x - nx, 0, pad_t - nt))
    cond = init_coef.unsqueeze(2).expand(b, init_coef.shape[1], pad_t // 4, nx, nx).reshape(b, -1, nx, nx)
    cond = F.pad(cond, (0, pad_x - nx, 0, pad_x - nx))
    so = smokeout.permute(0, 2, 1)                                         # [B, t, 2]
    so = so.reshape(b, 1, nt, 2, 1, 1).expand(b, 1, nt, 2, pad_x // 2, pad_x).reshape(b, 1, nt, pad_x, pad_x)
    so = F.pad(so, (0, 0, 0, 0, 0, pad_t - nt))
    state = torch.cat((data, cond.unsqueeze(1), so), dim=1)
    return (state.permute(0, 2, 1, 3, 4) / rescaler).contiguous()


def pack_smoke_gpu(coef, init_coef, smokeout, rescaler, idx=None, pad_t=24, pad_x=40):
    """wdno_pack_smoke_state (csrc/pack.hip): states [B, pad_t, 8 F + 2, pad_x, pad_x] / rescaler from device-resident coefficient arrays in one
    launch. coef [N, F, 8, nt, nx, nx], init_coef [N, 4, nx, nx] or [N, 5, 4, nx, nx] as the offline transform stores it (field 0 is used),
    smokeout [N, 2, nt]; idx (int64 device tensor [B]) picks the simulations of the batch from the N resident ones (None: all of them, in order)."""
    from wdno_amd import _lib
    from wdno_amd.ops import _lib_, _p, _stream
    for t, name in ((coef, 'coef'), (init_coef, 'init_coef'), (smokeout, 'smokeout')):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f'wdno_amd pack_smoke_gpu: {name} must be a contiguous float32 tensor on the GPU')
    n, nf = coef.shape[0], coef.shape[1]
    nt, nx = coef.shape[-3], coef.shape[-1]
    assert tuple(coef.shape[2:]) == (8, nt, nx, nx) and tuple(init_coef.shape[-3:]) == (4, nx, nx) and tuple(smokeout.shape[1:]) == (2, nt)
    c = 8 * nf + 2
    r = torch.as_tensor(rescaler, dtype=torch.float32).reshape(-1).to(coef.device)
    assert r.numel() == c, f'rescaler has {r.numel()} entries, the state {c} channels'
    if idx is not None:
        idx = idx.to(device=coef.device, dtype=torch.int64).contiguous()
    b = n if idx is None else idx.numel()
    out = torch.empty((b, pad_t, c, pad_x, pad_x), device=coef.device, dtype=torch.float32)
    _lib.check(_lib_().wdno_pack_smoke_state(_p(coef), coef[0].numel(), _p(init_coef), init_coef[0].numel(), _p(smokeout), smokeout[0].numel(), _p(idx), _p(r),
                                             _p(out), b, nf, nt, nx, pad_t, pad_x, _stream()), 'pack_smoke_state')
    return out


def pack_smoke_fields(fields, curve, rescaler, wave='bior1.3', pad_t=24, pad_x=40, direct=True):
    """fields [B, F, T, H, W] (field 0 = density) and the smoke-out curve [B, T], resident in HBM -> states [B, pad_t, 8 F + 2, pad_x, pad_x] /
    rescaler in TWO launches: the fused 3-D analysis of all fields whose store lands in the state (sub-band -> channel, frame -> frame, already
    divided: wdno_dwt_fwd_packed, csrc/dwt.hip) and wdno_pack_smoke_fields(coef = NULL) (csrc/pack.hip), which writes the zero padding and
    transforms the two condition channels -- the 2-D DWT of rho(t = 0) and the 1-D DWT of the curve, zero mode -- on the fly. direct=False (and
    whatever the direct store does not take): the analysis into a coefficient tensor, then the packing launch reads it (round 6's first form: the
    same bits, 26.6 MB written and read once more). The online form of wave_trans_2d.transform_simulation + Smoke_wave.__getitem__
    (smoke/wave_trans_2d.py:150-170, data_2d.py:156-221); equal to that chain to fp32 rounding (tests/test_gpu_data.py)."""
    import ctypes as C
    from wdno_amd import _lib, wavelets as Wv
    from wdno_amd.filters import filter_bank
    from wdno_amd.ops import _dwt_desc, _lib_, _p, _stream
    if not (fields.is_cuda and fields.dtype == torch.float32 and fields.is_contiguous() and curve.is_cuda and curve.dtype == torch.float32 and curve.is_contiguous()):
        raise RuntimeError('wdno_amd pack_smoke_fields: fields and curve must be contiguous float32 tensors on the GPU')
    b, nf, t0, h0, w0 = fields.shape
    dl, dh, _, _ = filter_bank(wave)
    L = len(dl)
    nt, nx = (t0 + L - 1) // 2, (w0 + L - 1) // 2
    c = 8 * nf + 2
    r = torch.as_tensor(rescaler, dtype=torch.float32).reshape(-1).to(fields.device)
    assert r.numel() == c and tuple(curve.shape) == (b, t0)
    out = torch.empty((b, pad_t, c, pad_x, pad_x), device=fields.device, dtype=torch.float32)
    fl, fh = (C.c_float * L)(*[float(v) for v in dl]), (C.c_float * L)(*[float(v) for v in dh])
    lib = _lib_()
    coef = None
    if direct and h0 == w0 and nt <= pad_t and nx <= pad_x:
        filt, _ = Wv._filters(wave)
        px2 = pad_x * pad_x
        d = _dwt_desc(3, Wv.MODES['zero'], L, b * nf, [t0, h0, w0], [nt, (h0 + L - 1) // 2, nx], (8 * px2, px2, c * px2, pad_x))
        rc = lib.wdno_dwt_fwd_packed(_p(fields), _p(out), C.byref(d), (C.c_float * len(filt))(*filt), nf, pad_t * c * px2, pad_x, _p(r), _stream())
        if rc == -3:                                   # WDNO_EUNSUPPORTED: the fused kernel does not take this shape
            direct = False
        else:
            _lib.check(rc, 'dwt_fwd_packed')
    else:
        direct = False
    if not direct:
        coef = Wv.dwt_packed(fields.reshape(b * nf, t0, h0, w0), wave, 'zero', 3)             # [B F, 8, nt, nx, nx]
        assert coef.shape[-3] == nt and coef.shape[-1] == nx
    _lib.check(lib.wdno_pack_smoke_fields(_p(coef), nf * 8 * nt * nx * nx, _p(fields), nf * t0 * h0 * w0, _p(curve), t0, fl, fh, L, _p(r), _p(out),
                                          b, nf, nt, nx, pad_t, pad_x, h0, w0, t0, _stream()), 'pack_smoke_fields')
    return out


class Smoke_wave(Dataset):
    def __init__(self, dataset_path, wave_type, pad_mode, is_train=True, is_super_model=False, downsample_type='time', N_downsample=0):
        super().__init__()
        assert is_train
        if wave_type not in _RESCALERS:
            raise ValueError(f'no RESCALER for wavelet {wave_type!r}')
        self.root, self.wave_type, self.pad_mode = dataset_path, wave_type, pad_mode
        self.is_train, self.is_super_model = is_train, is_super_model
        self.dirname = 'train'
        self.downsample_type = downsample_type
        self.N_downsample = N_downsample if is_super_model else 0
        self.n_simu = 20000
        r = torch.tensor(_RESCALERS[wave_type]).reshape(1, 42, 1, 1)
        if is_super_model:                                   # coefficient channels twice (fine, coarse), conditions once
            r = torch.cat((r[:, :40].repeat(1, 2, 1, 1), r[:, -2:]), dim=1)
        self.RESCALER = r

    def __len__(self):
        return self.n_simu

    def path(self, sim_id):
        return os.path.join(self.root, self.dirname, f'{self.wave_type}_{self.pad_mode}', f'{self.downsample_type}_downsample', f'{sim_id:06d}')

    def raw(self, sim_id):
        """The arrays of simulation sim_id as the offline transform stored them, for the base-resolution model: (coef [5, 8, nt, nx, nx],
        init_coef [4, nx, nx] of the density, smokeout [2, nt]) -- what wdno_amd.loader.ResidentSmokeLoader keeps in HBM and packs on the GPU."""
        db = torch.load(self.path(sim_id), weights_only=False)
        lvl = self.N_downsample
        return (db['coef'][lvl].to(torch.float32).contiguous(), db['init_coef'][lvl][0].to(torch.float32).contiguous(),
                db['smokeout'][lvl].to(torch.float32).contiguous())

    def __getitem__(self, sim_id):
        db = torch.load(self.path(sim_id), weights_only=False)
        lvl = self.N_downsample
        coef = db['coef'][lvl]
        ori_shape = list(db['ori_shape'])
        if self.downsample_type == 'time':
            ori_shape[0] = math.ceil(ori_shape[0] / 2 ** lvl)
            pad_t, pad_x = int(24 / 2 ** lvl), 40
        else:
            ori_shape[1] = math.ceil(ori_shape[1] / 2 ** lvl)
            ori_shape[2] = math.ceil(ori_shape[2] / 2 ** lvl)
            pad_t, pad_x = 24, int(40 / 2 ** lvl)
        state = pack_smoke_state(coef, db['init_coef'][lvl], db['smokeout'][lvl], self.RESCALER,
                                 coef_sub=db['coef'][lvl + 1] if self.is_super_model else None,
                                 downsample_type=self.downsample_type, pad_t=pad_t, pad_x=pad_x)
        return state, list(coef.shape[2:]), ori_shape, sim_id


class SuperDataLoader:
    """One batch per iteration, drawn from a randomly chosen dataset of the list (super-resolution levels)."""

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


_reference_getattr = wdno_amd.reference_fallthrough('ddpm.data_2d', __file__)


def __getattr__(name):           # `Smoke` (raw fields, not on the WDNO path) and anything else come from the reference module
    return _reference_getattr(name)
