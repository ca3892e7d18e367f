"""Golden vectors for the smoke control objective by RUNNING the reference's guidance_fn.

Build-container only (needs /root/reference):   python tests/golden/make_ref_guidance_golden.py
Writes tests/golden/ref_guidance.npz -- data only: the gradients smoke/inference_2d.py:30-66 returns for inputs that
tests/helpers.guidance_input() regenerates exactly on any host (an integer hash, no random generator), so only the outputs
are stored, and of those only the entries that can be non-zero (coefficient block + smoke-out channel) plus the maximum of the rest.

What runs is the reference's own code: inference_2d.guidance_fn, wave_trans_2d.tensor_to_coef and torch autograd. The reference
calls three third-party routines that are absent here (ptwt.waverec3, pytorch_wavelets.DWT1DInverse, pywt.Wavelet); differentiable
torch stand-ins for them are installed under those names and CHECKED below against oracle/dwt_ref.py (itself pinned against
PyWavelets 1.1.1 outputs, tests/golden/dwt_pywt.npz) to 1e-12 in fp64 before anything is generated. dataset/evaluate_solver.py
(PhiFlow solver evaluation, out of scope) is shadowed by an empty module: guidance_fn never touches it.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_ref_golden as M  # noqa: E402
from oracle import dwt_ref as R  # noqa: E402
from tests.helpers import guidance_input  # noqa: E402

M.STUBS['matplotlib/__init__.py'] = ''
M.STUBS['matplotlib/pyplot.py'] = ''
M.STUBS['dataset/__init__.py'] = ''
M.STUBS['dataset/evaluate_solver.py'] = ''
M.install_stubs()


# ---- differentiable stand-ins (zero-padding mode synthesis = transposed stride-2 convolution, then drop p = (2L-3)//2 at both ends)
def _synth(lo, hi, wave, axis):
    _, _, gl, gh = R.filter_bank(wave)
    lo, hi = lo.movedim(axis, -1), hi.movedim(axis, -1)
    m_, L = lo.shape[-1], len(gl)

    def up(t):
        z = torch.zeros(t.shape[:-1] + (2 * m_ - 1,), dtype=t.dtype)
        z[..., ::2] = t
        return z
    ul, uh = up(lo), up(hi)
    y = 0
    for m in range(L):
        y = y + F.pad(ul * float(gl[m]) + uh * float(gh[m]), (m, L - 1 - m))
    p = (2 * L - 3) // 2
    return y[..., p:y.shape[-1] - p].movedim(-1, axis)


class _Wavelet:
    def __init__(self, name):
        self.name = name


def _waverec3(coeffs, wavelet):
    c = dict(coeffs[1])
    c['aaa'] = coeffs[0]
    w = wavelet.name
    lvl = {th: _synth(c[th + 'a'], c[th + 'd'], w, -1) for th in ('aa', 'ad', 'da', 'dd')}
    lvl2 = {t: _synth(lvl[t + 'a'], lvl[t + 'd'], w, -2) for t in ('a', 'd')}
    return _synth(lvl2['a'], lvl2['d'], w, -3)


class _DWT1DInverse:
    def __init__(self, mode='zero', wave='db1'):
        assert mode == 'zero'
        self.wave = wave

    def to(self, device):
        return self

    def __call__(self, coeffs):
        lo, his = coeffs
        return _synth(lo, his[0], self.wave, -1)


# check the stand-ins against the pinned oracle before trusting them
_rng = np.random.default_rng(3)
_lll = _rng.standard_normal((3, 10, 18, 18))
_det = {k: _rng.standard_normal((3, 10, 18, 18)) for k in ('aad', 'ada', 'add', 'daa', 'dad', 'dda', 'ddd')}
_a = _waverec3([torch.from_numpy(_lll), {k: torch.from_numpy(v) for k, v in _det.items()}], _Wavelet('bior1.3')).numpy()
_b = R.idwt3(_lll, _det, 'bior1.3')
assert _a.shape == _b.shape and np.abs(_a - _b).max() < 1e-12, np.abs(_a - _b).max()
_lo, _hi = _rng.standard_normal((2, 1, 18)), _rng.standard_normal((2, 1, 18))
assert np.abs(_DWT1DInverse('zero', 'bior1.3')((torch.from_numpy(_lo), [torch.from_numpy(_hi)])).numpy() - R.idwt1d(_lo, _hi, 'bior1.3', 'zero')).max() < 1e-12

pw = types.ModuleType('pytorch_wavelets')
pw.DWT1DInverse = _DWT1DInverse
pw.DWTForward = pw.DWTInverse = pw.DWT1DForward = None
sys.modules['pytorch_wavelets'] = pw
pt = types.ModuleType('ptwt')
pt.waverec3 = _waverec3
sys.modules['ptwt'] = pt
py = types.ModuleType('pywt')
py.Wavelet = _Wavelet
sys.modules['pywt'] = py

with M.cuda_default_args_on_cpu():
    import inference_2d as INF                     # noqa: E402  /root/reference/smoke/inference_2d.py

CASES = {
    # name: (tensor shape, coefficient shape, field shape, is_condition_control, w_energy, w_init)
    'full': ((1, 24, 42, 40, 40), (18, 34, 34), (32, 64, 64), False, 0.7, 1.3),
    'full_control': ((1, 24, 42, 40, 40), (18, 34, 34), (32, 64, 64), True, 0.7, 1.3),
    'small_b2': ((2, 12, 42, 40, 40), (10, 18, 18), (16, 32, 32), False, 0.05, 0.4),
    'small_no_weights': ((2, 12, 42, 40, 40), (10, 18, 18), (16, 32, 32), False, 0.0, 0.0),
}

d = {}
for name, (tshape, shape, ori, control, w_e, w_i) in CASES.items():
    x, resc, init_u = guidance_input(tshape, shape, ori, seed=sum(name.encode()))
    args = types.SimpleNamespace(is_wavelet=True, wave_type='bior1.3', pad_mode='zero', is_condition_control=control)
    xr = x.clone().requires_grad_(True)
    # the sampler hands design_fn a tensor that requires grad (diffusion_2d.py:733-741); guidance_fn differentiates w.r.t. x * RESCALER
    g = INF.guidance_fn(xr, args, list(shape), list(ori), resc, w_energy=w_e, w_init=w_i, low=None, init=None, init_u=init_u)
    g = g.detach()
    tc, hc, wc = shape
    d[f'{name}::g_coef'] = g[:, :tc, :40, :hc, :wc].numpy().copy()
    d[f'{name}::g_smokeout'] = g[:, :tc, -1].numpy().copy()
    rest = g.clone()
    rest[:, :tc, :40, :hc, :wc] = 0
    rest[:, :tc, -1] = 0
    d[f'{name}::g_rest_absmax'] = np.array(float(rest.abs().max()))
    d[f'{name}::meta'] = np.array([*tshape, *shape, *ori, int(control)], dtype=np.int64)
    d[f'{name}::weights'] = np.array([w_e, w_i], dtype=np.float64)
    print(name, 'g norm', float(g.double().norm()), 'rest', float(rest.abs().max()))

out = os.path.join(HERE, 'ref_guidance.npz')
np.savez_compressed(out, **d)
print('wrote', out, os.path.getsize(out))
