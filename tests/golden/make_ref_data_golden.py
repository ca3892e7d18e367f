"""Golden vectors for the dataset-packing rows by importing the reference.

Build-container only (needs /root/reference):   python tests/golden/make_ref_data_golden.py
Writes tests/golden/ref_data_smoke.npz and ref_data_burgers.npz -- data only (synthetic coefficient files and what the
reference's Smoke_wave.__getitem__ / get_wavelet_super_preprocess make of them).

The reference calls pytorch_wavelets for the u0 / uT condition rows of the Burgers packer; that package is absent, so
stand-in modules backed by oracle/dwt_ref.py (itself pinned against PyWavelets) are installed under its name. Everything
else that runs is the reference's own code.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_ref_golden as M  # noqa: E402
from oracle import dwt_ref as R  # noqa: E402

M.STUBS['matplotlib/__init__.py'] = ''
M.STUBS['matplotlib/pyplot.py'] = ''
M.install_stubs()

# ---- oracle-backed stand-ins for the pytorch_wavelets classes the reference packer instantiates
pw = types.ModuleType('pytorch_wavelets')


class _Mod:
    def to(self, device):
        return self


class DWTInverse(_Mod):
    def __init__(self, mode='zero', wave='db1'):
        self.mode, self.wave = mode, wave

    def __call__(self, coeffs):
        yl, yh = coeffs
        return torch.from_numpy(R.waverec2(yl.double().numpy(), [h.double().numpy() for h in yh], self.wave, self.mode)).float()


class DWT1DForward(_Mod):
    def __init__(self, J=1, mode='zero', wave='db1'):
        self.J, self.mode, self.wave = J, mode, wave

    def __call__(self, x):
        lo, his = R.wavedec1(x.double().numpy(), self.wave, self.mode, self.J)
        return torch.from_numpy(lo).float(), [torch.from_numpy(h).float() for h in his]


pw.DWTInverse, pw.DWT1DForward = DWTInverse, DWT1DForward
pw.DWTForward = pw.DWT1DInverse = None
sys.modules['pytorch_wavelets'] = pw

with M.cuda_default_args_on_cpu():
    from ddpm.data_2d import Smoke_wave                                   # noqa: E402  smoke/ddpm/data_2d.py
    from ddpm_burgers import data_burgers_1d as DB                        # noqa: E402

rng = np.random.default_rng(77)

# =========================================================================================== smoke: per-simulation files
# small synthetic "simulation": fields [5, 8, 16, 16], smoke-out fraction [8]; file format of wave_trans_2d.py:172-185
X = rng.standard_normal((5, 8, 16, 16))
s_out = rng.random(8)
d = {}


def smoke_file(kind):
    coef, init, so = [], [], []
    for i in range(2):
        xs = X[:, ::2 ** i] if kind == 'time' else X[:, :, ::2 ** i, ::2 ** i]
        lll, det = R.dwt3(xs, 'bior1.3')
        coef.append(torch.from_numpy(R.smoke_coef_to_tensor(lll, det)).float())
        yl, yh = R.dwt2(xs[:, None, 0], 'bior1.3', 'zero')
        init.append(torch.from_numpy(np.concatenate([yl, yh[:, 0]], axis=1)).float())
        ss = s_out.reshape(1, 1, -1)[:, :, ::2 ** i] if kind == 'time' else s_out.reshape(1, 1, -1)
        lo, hi = R.dwt1d(ss, 'bior1.3', 'zero')
        so.append(torch.from_numpy(np.concatenate([lo, hi], axis=1)[0]).float())
    return {'coef': coef, 'init_coef': init, 'smokeout': so, 'shape': [c.shape[-3:] for c in coef], 'ori_shape': torch.Size(X.shape[1:])}


tmp = tempfile.mkdtemp(prefix='wdno_smoke_')
for kind in ('time', 'space'):
    os.makedirs(os.path.join(tmp, 'train', 'bior1.3_zero', f'{kind}_downsample'))
    f = smoke_file(kind)
    torch.save(f, os.path.join(tmp, 'train', 'bior1.3_zero', f'{kind}_downsample', '000000'))
    for i in range(2):
        d[f'{kind}_coef{i}'] = f['coef'][i].numpy()
        d[f'{kind}_init{i}'] = f['init_coef'][i].numpy()
        d[f'{kind}_smokeout{i}'] = f['smokeout'][i].numpy()
    d[f'{kind}_ori_shape'] = np.array(f['ori_shape'])
for tag, kw in (('base', dict(is_super_model=False)), ('super_time', dict(is_super_model=True, downsample_type='time', N_downsample=0)),
                ('super_space', dict(is_super_model=True, downsample_type='space', N_downsample=0))):
    ds = Smoke_wave(tmp, 'bior1.3', 'zero', **kw)
    state, shape, ori_shape, sim_id = ds[0]
    d[f'out_{tag}_state'] = state.numpy()
    d[f'out_{tag}_shape'] = np.array(shape)
    d[f'out_{tag}_ori_shape'] = np.array(ori_shape)
    d[f'out_{tag}_rescaler'] = ds.RESCALER.numpy()
np.savez_compressed(os.path.join(HERE, 'ref_data_smoke.npz'), **d)

# =========================================================================================== burgers: coefficient dataset
# file format of burgers/wave_trans.py:120-123: {'coef': [4 x [N, 2, 4, h, w]], 'shape', 'ori_shape'}
N = 3
U = rng.standard_normal((N, 2, 81, 120))
U[:, 1, -1] = 0                                   # f has 80 rows, zero-filled to 81 (wave_trans.py:74)
coef = []
for i in range(4):
    xs = U[:, :, ::2 ** i, ::2 ** i]
    yl, yh = R.dwt2(xs, 'bior2.4', 'periodization')
    coef.append(torch.from_numpy(R.burgers_coef_to_tensor(yl, yh)).float())
db = {'coef': coef, 'shape': [c.shape[2:] for c in coef], 'ori_shape': torch.Size((81, 120))}
b = {f'coef{i}': coef[i].numpy() for i in range(4)}
resc = torch.tensor([10, 3, 3, 1, 21, 5, 5, 1, 10]).view(1, 9, 1, 1).float()
for tag, kw in (('base', dict(rescaler=resc)), ('base_u0only', dict(rescaler=resc, is_condition_uT=False)),
                ('super0', dict(rescaler=torch.cat((resc[:, :8].repeat(1, 2, 1, 1), resc[:, 8:]), dim=1), is_super_model=True, N_downsample=0)),
                ('super1', dict(rescaler=torch.cat((resc[:, :8].repeat(1, 2, 1, 1), resc[:, 8:]), dim=1), is_super_model=True, N_downsample=1))):
    pre = DB.get_wavelet_super_preprocess(mode='periodization', wave_type='bior2.4', **kw)
    data, shape, ori_shape = pre({'coef': [c.clone() for c in coef], 'shape': db['shape'], 'ori_shape': db['ori_shape']})
    b[f'out_{tag}_data'] = data.numpy()
    b[f'out_{tag}_shape'] = np.array(shape)
    b[f'out_{tag}_ori_shape'] = np.array(ori_shape)
np.savez_compressed(os.path.join(HERE, 'ref_data_burgers.npz'), **b)
for fn in ('ref_data_smoke.npz', 'ref_data_burgers.npz'):
    print(fn, os.path.getsize(os.path.join(HERE, fn)))
