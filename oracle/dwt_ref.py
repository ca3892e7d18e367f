"""ORACLE (test infrastructure only) -- CPU restatement of the single-level wavelet transforms on the WDNO hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file. The product
path (wdno_amd/) never does.

What is restated
----------------
The reference does not contain the DWT arithmetic; it calls three un-vendored third-party packages
(env.sh:6-11): `pytorch_wavelets` (git HEAD, unpinned), `ptwt==0.1.6`, `PyWavelets` (unpinned). The
call sites this file stands in for are

  * pytorch_wavelets.DWTForward / DWTInverse (J=1, mode='periodization', 'bior2.4')
        burgers/wave_trans.py:94-98,107 ; burgers/ddpm_burgers/data_burgers_1d.py:66-68
  * pytorch_wavelets.DWT1DForward / DWT1DInverse
        burgers/ddpm_burgers/data_burgers_1d.py:72-73 ; smoke/wave_trans_2d.py:80,141
  * pytorch_wavelets.DWTForward (J=1, mode='zero', 'bior1.3')
        smoke/wave_trans_2d.py:79,136
  * ptwt.wavedec3 / ptwt.waverec3 (mode='zero', level=1, 'bior1.3')
        smoke/wave_trans_2d.py:129-132 ; smoke/inference_2d.py:41,141,184,250
  * the packing helpers coef_to_tensor / tensor_to_coef
        burgers/wave_trans.py:18-62 ; smoke/wave_trans_2d.py:17-58
  * upsample_coef   burgers/ddpm_burgers/wavelet_utils.py:5-16 ; smoke/ddpm/wave_utils.py:1-14

Parity pin: tests/golden/dwt_pywt.npz holds outputs of PyWavelets 1.1.1 (the library all three
wrappers test themselves against) produced by tests/golden/make_dwt_golden.py; tests/test_oracle_dwt.py
checks every function below against them in float64 (<= 1e-12).

Everything is written as a tensor product of four 1-D filter-bank primitives along the last axes.
All arithmetic is done in the dtype of the input (float64 for pinning, float32 for parity with HIP).
"""
import numpy as np

from wdno_amd.filters import filter_bank


# --------------------------------------------------------------------------- 1-D primitives
def _move(x, axis):
    return np.moveaxis(x, axis, -1)


def analysis_per(x, h, axis=-1):
    """One analysis branch, periodization mode (pytorch_wavelets 'periodization' == pywt 'periodization').

    If N is odd the last sample is repeated. x~[n] = x[(n + L/2) mod N];
    y[k] = sum_m h[L-1-m] * x~z[2k + m - (L-1)] (zero-extended), k = 0 .. N/2 + L/2 - 1;
    fold y[k] += y[k + N/2] for k < L/2 ; keep N/2 outputs.
    """
    x = _move(np.asarray(x), axis)
    h = np.asarray(h, dtype=x.dtype)
    L = h.shape[0]
    if x.shape[-1] % 2 == 1:
        x = np.concatenate([x, x[..., -1:]], axis=-1)
    N = x.shape[-1]
    xt = np.roll(x, -(L // 2), axis=-1)
    nk = N // 2 + L // 2
    xz = np.concatenate([np.zeros(x.shape[:-1] + (L - 1,), x.dtype), xt,
                         np.zeros(x.shape[:-1] + (L + 1,), x.dtype)], axis=-1)
    y = np.zeros(x.shape[:-1] + (nk,), x.dtype)
    for m in range(L):
        # x~z index 2k + m - (L-1)  ->  padded index 2k + m
        y += h[L - 1 - m] * xz[..., m:m + 2 * nk:2]
    out = y[..., :N // 2].copy()
    out[..., :L // 2] += y[..., N // 2:N // 2 + L // 2]
    return np.moveaxis(out, -1, axis)


def synthesis_per(lo, hi, g_lo, g_hi, axis=-1):
    """Synthesis, periodization mode. Output length is 2*M (always even)."""
    lo = _move(np.asarray(lo), axis)
    hi = _move(np.asarray(hi), axis)
    g_lo = np.asarray(g_lo, dtype=lo.dtype)
    g_hi = np.asarray(g_hi, dtype=lo.dtype)
    L = g_lo.shape[0]
    M = lo.shape[-1]
    N = 2 * M
    y = np.zeros(lo.shape[:-1] + (2 * M - 2 + L,), lo.dtype)
    for m in range(L):
        y[..., m:m + 2 * M:2] += lo * g_lo[m] + hi * g_hi[m]
    out = y[..., :N].copy()
    tail = y[..., N:]
    out[..., :tail.shape[-1]] += tail
    out = np.roll(out, 1 - L // 2, axis=-1)
    return np.moveaxis(out, -1, axis)


def analysis_zero(x, h, axis=-1):
    """One analysis branch, zero-padding mode (ptwt 'zero' / pytorch_wavelets 'zero').

    p = (2L-3)//2 zeros left, p (+1 if N odd) right; y[k] = sum_m h[L-1-m] xp[2k+m], k < (N+L-1)//2.
    """
    x = _move(np.asarray(x), axis)
    h = np.asarray(h, dtype=x.dtype)
    L = h.shape[0]
    N = x.shape[-1]
    p = (2 * L - 3) // 2
    nk = (N + L - 1) // 2
    xp = np.concatenate([np.zeros(x.shape[:-1] + (p,), x.dtype), x,
                         np.zeros(x.shape[:-1] + (p + (N % 2) + 2,), x.dtype)], axis=-1)
    y = np.zeros(x.shape[:-1] + (nk,), x.dtype)
    for m in range(L):
        y += h[L - 1 - m] * xp[..., m:m + 2 * nk:2]
    return np.moveaxis(y, -1, axis)


def synthesis_zero(lo, hi, g_lo, g_hi, axis=-1):
    """Synthesis, zero mode: transposed stride-2 convolution, then drop p=(2L-3)//2 from both ends."""
    lo = _move(np.asarray(lo), axis)
    hi = _move(np.asarray(hi), axis)
    g_lo = np.asarray(g_lo, dtype=lo.dtype)
    g_hi = np.asarray(g_hi, dtype=lo.dtype)
    L = g_lo.shape[0]
    M = lo.shape[-1]
    y = np.zeros(lo.shape[:-1] + (2 * M - 2 + L,), lo.dtype)
    for m in range(L):
        y[..., m:m + 2 * M:2] += lo * g_lo[m] + hi * g_hi[m]
    p = (2 * L - 3) // 2
    out = y[..., p:y.shape[-1] - p]
    return np.moveaxis(out, -1, axis)


_ANALYSIS = {'periodization': analysis_per, 'zero': analysis_zero}
_SYNTHESIS = {'periodization': synthesis_per, 'zero': synthesis_zero}


# --------------------------------------------------------------------------- 1-D transforms (DWT1DForward/Inverse)
def dwt1d(x, wave, mode):
    """x [N, C, L] -> (lo [N,C,L'], hi [N,C,L'])   (pytorch_wavelets.DWT1DForward(J=1): returns (x0, [x1]))."""
    dl, dh, _, _ = filter_bank(wave)
    a = _ANALYSIS[mode]
    return a(x, dl, -1), a(x, dh, -1)


def idwt1d(lo, hi, wave, mode):
    _, _, rl, rh = filter_bank(wave)
    return _SYNTHESIS[mode](lo, hi, rl, rh, -1)


# --------------------------------------------------------------------------- 2-D transforms (DWTForward/Inverse, J=1)
def dwt2(x, wave, mode):
    """x [N, C, H, W] -> (Yl [N,C,H',W'], Yh [N,C,3,H',W']).

    Filters run along W first, then H (pytorch_wavelets lowlevel afb2d). Band order in Yh:
      0: lo along W, hi along H   (pywt cH, 'da')
      1: hi along W, lo along H   (pywt cV, 'ad')
      2: hi along W, hi along H   (pywt cD, 'dd')
    """
    dl, dh, _, _ = filter_bank(wave)
    a = _ANALYSIS[mode]
    lo_w = a(x, dl, -1)
    hi_w = a(x, dh, -1)
    ll = a(lo_w, dl, -2)
    lh = a(lo_w, dh, -2)
    hl = a(hi_w, dl, -2)
    hh = a(hi_w, dh, -2)
    return ll, np.stack([lh, hl, hh], axis=2)


def idwt2(yl, yh, wave, mode):
    """Inverse of dwt2: combine along H first, then along W (pytorch_wavelets sfb2d)."""
    _, _, rl, rh = filter_bank(wave)
    s = _SYNTHESIS[mode]
    lh, hl, hh = yh[:, :, 0], yh[:, :, 1], yh[:, :, 2]
    lo_w = s(yl, lh, rl, rh, -2)
    hi_w = s(hl, hh, rl, rh, -2)
    return s(lo_w, hi_w, rl, rh, -1)


# --------------------------------------------------------------------------- 3-D transforms (ptwt.wavedec3 / waverec3, level=1)
BANDS3 = ('aaa', 'aad', 'ada', 'add', 'daa', 'dad', 'dda', 'ddd')   # letter order = axis order (T, H, W)


def dwt3(x, wave, mode='zero'):
    """x [N, T, H, W] -> (lll [N,T',H',W'], dict of the 7 detail bands in BANDS3[1:] order).

    ptwt applies one conv3d with the 8 outer-product kernels; separable application is the same operator.
    """
    dl, dh, _, _ = filter_bank(wave)
    a = _ANALYSIS[mode]
    out = {}
    for key in BANDS3:
        y = x
        for ax, letter in zip((-3, -2, -1), key):
            y = a(y, dl if letter == 'a' else dh, ax)
        out[key] = y
    lll = out.pop('aaa')
    return lll, out


def idwt3(lll, details, wave, mode='zero'):
    """Inverse of dwt3 -> [N, 2T'-2+L-2p, ...] (18,34,34 -> 32,64,64 for bior1.3)."""
    _, _, rl, rh = filter_bank(wave)
    s = _SYNTHESIS[mode]
    c = dict(details)
    c['aaa'] = lll
    # combine along W, then H, then T
    lvl = {}
    for th in ('aa', 'ad', 'da', 'dd'):
        lvl[th] = s(c[th + 'a'], c[th + 'd'], rl, rh, -1)
    lvl2 = {}
    for t in ('a', 'd'):
        lvl2[t] = s(lvl[t + 'a'], lvl[t + 'd'], rl, rh, -2)
    return s(lvl2['a'], lvl2['d'], rl, rh, -3)


# --------------------------------------------------------------------------- multi-level (J > 1) transforms
def _crop_like(ll, ref_shape, naxes):
    """pytorch_wavelets.DWTInverse / pywt.waverec*: an approximation one sample longer than the next detail band
    (odd length one level up) loses its last sample before the synthesis."""
    sl = [slice(None)] * ll.ndim
    for ax in range(-naxes, 0):
        if ll.shape[ax] == ref_shape[ax] + 1:
            sl[ax] = slice(0, -1)
    return ll[tuple(sl)]


def wavedec1(x, wave, mode, J):
    """pytorch_wavelets.DWT1DForward(J): (x0, [x1 finest, ..., coarsest])."""
    highs = []
    for _ in range(J):
        x, hi = dwt1d(x, wave, mode)
        highs.append(hi)
    return x, highs


def waverec1(lo, highs, wave, mode):
    for hi in highs[::-1]:
        lo = idwt1d(_crop_like(lo, hi.shape, 1), hi, wave, mode)
    return lo


def wavedec2(x, wave, mode, J):
    """pytorch_wavelets.DWTForward(J): (Yl, [Yh finest, ..., Yh coarsest]); each level re-analyses the LL band
    (burgers/wave_trans.py:94 builds J = dwt_max_level; data uses J = 1)."""
    yh = []
    for _ in range(J):
        x, h = dwt2(x, wave, mode)
        yh.append(h)
    return x, yh


def waverec2(yl, yh, wave, mode):
    for h in yh[::-1]:
        yl = idwt2(_crop_like(yl, h.shape, 2), h, wave, mode)
    return yl


def wavedec3(x, wave, mode='zero', level=1):
    """ptwt.wavedec3: [lll, {coarsest details}, ..., {finest details}]."""
    dicts = []
    for _ in range(level):
        x, det = dwt3(x, wave, mode)
        dicts.append(det)
    return [x] + dicts[::-1]


def waverec3(coeffs, wave, mode='zero'):
    lll = coeffs[0]
    for det in coeffs[1:]:
        lll = idwt3(_crop_like(lll, det['aad'].shape, 3), det, wave, mode)
    return lll


# --------------------------------------------------------------------------- packing helpers
def burgers_coef_to_tensor(yl, yh, pad=False):
    """burgers/wave_trans.py:43-62: J = 1 -> stack (Yl, Yh[0]) -> [N, C, 4, H', W'] (optionally zero-pad to 64x64
    multiples). J > 1 (yh a list, finest first): every band is nearest-repeated to the finest grid (Yl and Yh[i] by
    2^(J-1) / 2^i) and the row deficit of Yh[i] is filled by replicating its last row (wave_trans.py:50-57)."""
    if isinstance(yh, (list, tuple)) and len(yh) > 1:
        J = len(yh)
        rows = yh[0].shape[-2] + 2 ** (J - 1) - 1
        t = np.zeros((yl.shape[0], yl.shape[1], 1 + 3 * J, rows, yh[0].shape[-1]), dtype=yl.dtype)
        t[:, :, 0] = np.repeat(np.repeat(yl, 2 ** (J - 1), axis=-2), 2 ** (J - 1), axis=-1)
        for i in range(J):
            r = np.repeat(np.repeat(yh[i], 2 ** i, axis=-2), 2 ** i, axis=-1)
            fill = np.repeat(r[:, :, :, -1:], 2 ** (J - 1) - 2 ** i, axis=3)
            t[:, :, 1 + 3 * i:1 + 3 * (i + 1)] = np.concatenate([r, fill], axis=3)
    else:
        if isinstance(yh, (list, tuple)):
            yh = yh[0]
        t = np.concatenate([yl[:, :, None], yh], axis=2)
    if pad:
        up_t = int(t.shape[-2] / 40)
        up_x = int(t.shape[-1] / 60)
        t = np.pad(t, [(0, 0)] * 3 + [(0, 64 * up_t - t.shape[-2]), (0, 64 * up_x - t.shape[-1])])
    return t


def burgers_tensor_to_coef(coef_tensor, shape):
    """burgers/wave_trans.py:30-40: [N, >=8, Hp, Wp] -> (Yl [N,2,h,w], [Yh [N,2,3,h,w]]) with crop."""
    h, w = int(shape[-2]), int(shape[-1])
    u = coef_tensor[:, 0:4, :h, :w]
    f = coef_tensor[:, 4:8, :h, :w]
    yl = np.stack([u[:, 0], f[:, 0]], axis=1)
    yh = np.stack([u[:, 1:4], f[:, 1:4]], axis=1)
    return yl, yh


def smoke_coef_to_tensor(lll, details):
    """smoke/wave_trans_2d.py:55-58: [N, 8, T', H', W'] in BANDS3 order."""
    return np.stack([lll] + [details[k] for k in BANDS3[1:]], axis=1)


def smoke_tensor_to_coef(coef_tensor, shape, upsample_type=None, nfields=5):
    """smoke/wave_trans_2d.py:17-52: [N, >=8*nfields, Tp, Hp, Wp] -> (Yl [N*nfields,t,h,w], dict of 7)."""
    t, h, w = int(shape[-3]), int(shape[-2]), int(shape[-1])
    t0 = 1 if upsample_type == 'time' else 0
    s0 = 1 if upsample_type == 'space' else 0
    c = coef_tensor[:, :8 * nfields, t0:t0 + t, s0:s0 + h, s0:s0 + w]
    c = c.reshape(c.shape[0], nfields, 8, t, h, w).reshape(-1, 8, t, h, w)
    return c[:, 0], {k: c[:, i + 1] for i, k in enumerate(BANDS3[1:])}


def upsample_coef_2d(w_sub):
    """burgers/ddpm_burgers/wavelet_utils.py:5-16: nearest x2 along the last two axes of [N, l, nt, nx]."""
    return np.repeat(np.repeat(w_sub, 2, axis=-2), 2, axis=-1)


def upsample_coef_3d(w_sub, kind):
    """smoke/ddpm/wave_utils.py:1-14: [N, nt, l, nx, nx] -> x2 in time or in both space axes."""
    if kind == 'time':
        return np.repeat(w_sub, 2, axis=1)
    return np.repeat(np.repeat(w_sub, 2, axis=-2), 2, axis=-1)
