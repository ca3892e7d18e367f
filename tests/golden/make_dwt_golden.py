"""Generate tests/golden/dwt_pywt.npz from PyWavelets 1.1.1.

Run in the build container with the interpreter that has PyWavelets:
    /opt/conda/bin/python3.9 tests/golden/make_dwt_golden.py
The main interpreter (numpy 2.x, torch) only reads the resulting .npz. PyWavelets is the numerical
ground truth that pytorch_wavelets and ptwt (the libraries the reference calls) test themselves against.
"""
import os
import numpy as np
import pywt

assert pywt.__version__ == '1.1.1', pywt.__version__
rng = np.random.default_rng(20250928)
out = {}


def put(name, arr):
    out[name] = np.ascontiguousarray(arr, dtype=np.float64)


# ---- 2-D periodization (Burgers): pywt.dwt2 returns (cA, (cH, cV, cD))
for tag, wave, shape in [('ref', 'bior2.4', (1, 2, 81, 120)), ('small', 'bior2.4', (2, 2, 16, 24)),
                         ('db4', 'db4', (2, 1, 19, 22)), ('sym4', 'sym4', (1, 2, 20, 18)), ('haar', 'haar', (1, 1, 7, 10)),
                         ('b13', 'bior1.3', (1, 2, 12, 16))]:
    x = rng.standard_normal(shape)
    cA, (cH, cV, cD) = pywt.dwt2(x, wave, mode='periodization', axes=(-2, -1))
    put(f'dwt2per_{tag}_x', x)
    put(f'dwt2per_{tag}_yl', cA)
    put(f'dwt2per_{tag}_yh', np.stack([cH, cV, cD], axis=2))
    out[f'dwt2per_{tag}_wave'] = np.array(wave)
    # inverse on *random* coefficients (sampler-like input, not a consistent decomposition)
    r = [rng.standard_normal(cA.shape) for _ in range(4)]
    rec = pywt.idwt2((r[0], (r[1], r[2], r[3])), wave, mode='periodization', axes=(-2, -1))
    put(f'idwt2per_{tag}_yl', r[0])
    put(f'idwt2per_{tag}_yh', np.stack(r[1:], axis=2))
    put(f'idwt2per_{tag}_x', rec)

# ---- 1-D both modes
for tag, wave, mode, shape in [('per', 'bior2.4', 'periodization', (3, 2, 120)), ('perodd', 'bior2.4', 'periodization', (2, 1, 81)),
                               ('zero', 'bior1.3', 'zero', (1, 1, 32)), ('zeroodd', 'bior1.3', 'zero', (2, 2, 33)),
                               ('zdb4', 'db4', 'zero', (1, 2, 21))]:
    x = rng.standard_normal(shape)
    lo, hi = pywt.dwt(x, wave, mode=mode, axis=-1)
    put(f'dwt1_{tag}_x', x)
    put(f'dwt1_{tag}_lo', lo)
    put(f'dwt1_{tag}_hi', hi)
    out[f'dwt1_{tag}_wave'] = np.array(wave)
    out[f'dwt1_{tag}_mode'] = np.array(mode)
    rl, rh = rng.standard_normal(lo.shape), rng.standard_normal(hi.shape)
    put(f'idwt1_{tag}_lo', rl)
    put(f'idwt1_{tag}_hi', rh)
    put(f'idwt1_{tag}_x', pywt.idwt(rl, rh, wave, mode=mode, axis=-1))

# ---- 2-D zero mode (initial-density transform, smoke)
for tag, wave, shape in [('ref', 'bior1.3', (2, 1, 64, 64)), ('odd', 'bior1.3', (1, 2, 9, 11))]:
    x = rng.standard_normal(shape)
    cA, (cH, cV, cD) = pywt.dwt2(x, wave, mode='zero', axes=(-2, -1))
    put(f'dwt2zero_{tag}_x', x)
    put(f'dwt2zero_{tag}_yl', cA)
    put(f'dwt2zero_{tag}_yh', np.stack([cH, cV, cD], axis=2))
    out[f'dwt2zero_{tag}_wave'] = np.array(wave)

# ---- 3-D zero mode (smoke): pywt.dwtn keys 'aaa'..'ddd', letter order = axis order
KEYS = ('aaa', 'aad', 'ada', 'add', 'daa', 'dad', 'dda', 'ddd')
for tag, wave, shape in [('mid', 'bior1.3', (2, 16, 20, 24)), ('odd', 'bior1.3', (2, 9, 11, 13)), ('db2', 'db2', (1, 8, 10, 12))]:
    x = rng.standard_normal(shape)
    c = pywt.dwtn(x, wave, mode='zero', axes=(1, 2, 3))
    put(f'dwt3_{tag}_x', x)
    put(f'dwt3_{tag}_coef', np.stack([c[k] for k in KEYS], axis=1))
    out[f'dwt3_{tag}_wave'] = np.array(wave)
    r = {k: rng.standard_normal(c[k].shape) for k in KEYS}
    put(f'idwt3_{tag}_coef', np.stack([r[k] for k in KEYS], axis=1))
    put(f'idwt3_{tag}_x', pywt.idwtn(r, wave, mode='zero', axes=(1, 2, 3)))

# ---- multi-level (J > 1): pywt.wavedec2 / wavedec / wavedecn return coarsest first
for tag, wave, shape, J in [('ref3', 'bior2.4', (1, 2, 81, 120), 3), ('even2', 'bior2.4', (2, 1, 32, 48), 2), ('db4odd', 'db4', (1, 1, 37, 26), 2)]:
    x = rng.standard_normal(shape)
    c = pywt.wavedec2(x, wave, mode='periodization', level=J, axes=(-2, -1))
    put(f'wavedec2_{tag}_x', x)
    put(f'wavedec2_{tag}_yl', c[0])
    for lvl in range(J):                      # store finest first (pytorch_wavelets order)
        cH, cV, cD = c[J - lvl]
        put(f'wavedec2_{tag}_yh{lvl}', np.stack([cH, cV, cD], axis=2))
    out[f'wavedec2_{tag}_wave'] = np.array(wave)
    out[f'wavedec2_{tag}_J'] = np.array(J)
    r = [rng.standard_normal(c[0].shape)] + [tuple(rng.standard_normal(b.shape) for b in lv) for lv in c[1:]]
    put(f'waverec2_{tag}_yl', r[0])
    for lvl in range(J):
        put(f'waverec2_{tag}_yh{lvl}', np.stack(r[J - lvl], axis=2))
    put(f'waverec2_{tag}_x', pywt.waverec2(r, wave, mode='periodization', axes=(-2, -1)))

for tag, wave, mode, shape, J in [('per3', 'bior2.4', 'periodization', (2, 2, 81), 3), ('zero2', 'bior1.3', 'zero', (1, 2, 33), 2)]:
    x = rng.standard_normal(shape)
    c = pywt.wavedec(x, wave, mode=mode, level=J, axis=-1)
    put(f'wavedec1_{tag}_x', x)
    put(f'wavedec1_{tag}_lo', c[0])
    for lvl in range(J):
        put(f'wavedec1_{tag}_hi{lvl}', c[J - lvl])
    out[f'wavedec1_{tag}_wave'] = np.array(wave)
    out[f'wavedec1_{tag}_mode'] = np.array(mode)
    out[f'wavedec1_{tag}_J'] = np.array(J)
    put(f'waverec1_{tag}_x', pywt.waverec(c, wave, mode=mode, axis=-1))

for tag, wave, shape, J in [('l2', 'bior1.3', (2, 16, 20, 24), 2), ('l2odd', 'bior1.3', (1, 13, 18, 15), 2)]:
    x = rng.standard_normal(shape)
    c = pywt.wavedecn(x, wave, mode='zero', level=J, axes=(1, 2, 3))
    put(f'wavedec3_{tag}_x', x)
    put(f'wavedec3_{tag}_lll', c[0])
    for lvl in range(J):                      # ptwt order: coarsest first, like pywt
        put(f'wavedec3_{tag}_d{lvl}', np.stack([c[1 + lvl][k] for k in KEYS[1:]], axis=1))
    out[f'wavedec3_{tag}_wave'] = np.array(wave)
    out[f'wavedec3_{tag}_J'] = np.array(J)
    put(f'waverec3_{tag}_x', pywt.waverecn(c, wave, mode='zero', axes=(1, 2, 3)))

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dwt_pywt.npz')
np.savez_compressed(dst, **out)
print('wrote', dst, os.path.getsize(dst), 'bytes,', len(out), 'arrays')
