"""DWT / IDWT operators on MI355X (autograd-aware), behind the call signatures the reference uses.

The reference calls three third-party packages that are not dependencies of this repo:
  pytorch_wavelets.DWTForward / DWTInverse / DWT1DForward / DWT1DInverse   (burgers/wave_trans.py:94-98,
      burgers/ddpm_burgers/data_burgers_1d.py:66-73, smoke/wave_trans_2d.py:79-80, smoke/inference_2d.py:43,144)
  ptwt.wavedec3 / ptwt.waverec3 (level 1)                                   (smoke/wave_trans_2d.py:129-132)
  pywt.Wavelet(name)                                                        (filter taps only)
wdno_amd/third_party/ re-exports the classes / functions below under those module names.

Every transform level is one C-ABI call (wdno_dwt_fwd / wdno_dwt_inv) which writes / reads the sub-bands *already stacked*
in coef_to_tensor order, so `dwt2_packed` / `dwt3_packed` give the packed training tensor with no extra copies.
Gradients use the exact adjoint kernels (wdno_dwt_*_adjoint), which is what guidance back-propagation through the
IDWT needs in every sampling step (smoke/inference_2d.py:41,65).
"""
import torch
from torch import nn

from . import ops
from .filters import dwt_coeff_len, filter_bank

MODES = {'periodization': 0, 'per': 0, 'zero': 1}
BANDS3 = ('aaa', 'aad', 'ada', 'add', 'daa', 'dad', 'dda', 'ddd')


def _wave_name(wave):
    return wave if isinstance(wave, str) else getattr(wave, 'name')


def _filters(wave):
    dl, dh, rl, rh = filter_bank(_wave_name(wave))
    return [float(v) for v in (*dl, *dh, *rl, *rh)], len(dl)


def _dims3(sig_dims, nd):
    return [1] * (3 - nd) + list(sig_dims)


class _Analysis(torch.autograd.Function):
    """x [n_img, *sig] -> packed coefficients [n_img, 2^nd, *coef] (bands stacked in pywt order)."""

    @staticmethod
    def forward(ctx, x, wave, mode, nd):
        x = ops._chk(x, 'x')
        filt, L = _filters(wave)
        m = MODES[mode]
        sig = list(x.shape[-nd:])
        n_img = x.shape[0]
        cdims = [dwt_coeff_len(s, L, 'periodization' if m == 0 else 'zero') for s in sig]
        nb = 2 ** nd
        out = torch.empty((n_img, nb, *cdims), device=x.device, dtype=torch.float32)
        c3 = _dims3(cdims, nd)
        cs = (nb * c3[0] * c3[1] * c3[2], c3[0] * c3[1] * c3[2], c3[1] * c3[2], c3[2])
        ops.dwt_call('fwd', x, out, nd, m, filt, n_img, _dims3(sig, nd), c3, cs)
        ctx.meta = (wave, m, nd, sig, cdims, n_img, filt)
        return out

    @staticmethod
    def backward(ctx, g):
        wave, m, nd, sig, cdims, n_img, filt = ctx.meta
        g = ops._chk(g, 'grad')
        dx = torch.empty((n_img, *sig), device=g.device, dtype=torch.float32)
        c3 = _dims3(cdims, nd)
        nb = 2 ** nd
        cs = (nb * c3[0] * c3[1] * c3[2], c3[0] * c3[1] * c3[2], c3[1] * c3[2], c3[2])
        ops.dwt_call('fwd_adjoint', g, dx, nd, m, filt, n_img, _dims3(sig, nd), c3, cs)
        return dx, None, None, None


class _Synthesis(torch.autograd.Function):
    """packed coefficients [n_img, 2^nd, *coef] -> x [n_img, *sig] with the natural output size."""

    @staticmethod
    def forward(ctx, coef, wave, mode, nd):
        coef = ops._chk(coef, 'coef')
        filt, L = _filters(wave)
        m = MODES[mode]
        cdims = list(coef.shape[-nd:])
        n_img = coef.shape[0]
        sig = [2 * c if m == 0 else 2 * c - L + 2 for c in cdims]
        out = torch.empty((n_img, *sig), device=coef.device, dtype=torch.float32)
        c3 = _dims3(cdims, nd)
        nb = 2 ** nd
        cs = (nb * c3[0] * c3[1] * c3[2], c3[0] * c3[1] * c3[2], c3[1] * c3[2], c3[2])
        ops.dwt_call('inv', coef, out, nd, m, filt, n_img, _dims3(sig, nd), c3, cs)
        ctx.meta = (m, nd, sig, cdims, n_img, filt)
        return out

    @staticmethod
    def backward(ctx, g):
        m, nd, sig, cdims, n_img, filt = ctx.meta
        g = ops._chk(g, 'grad')
        nb = 2 ** nd
        dc = torch.empty((n_img, nb, *cdims), device=g.device, dtype=torch.float32)
        c3 = _dims3(cdims, nd)
        cs = (nb * c3[0] * c3[1] * c3[2], c3[0] * c3[1] * c3[2], c3[1] * c3[2], c3[2])
        ops.dwt_call('inv_adjoint', g, dc, nd, m, filt, n_img, _dims3(sig, nd), c3, cs)
        return dc, None, None, None


# ----------------------------------------------------------------------------------------------------- packed API
def dwt_packed(x, wave, mode, nd):
    """x [..., *sig(nd)] -> [..., 2^nd, *coef]: sub-bands stacked in coef_to_tensor order."""
    lead = x.shape[:-nd]
    out = _Analysis.apply(x.reshape(-1, *x.shape[-nd:]), wave, mode, nd)
    return out.reshape(*lead, *out.shape[1:])


def idwt_packed(coef, wave, mode, nd):
    """[..., 2^nd, *coef] -> [..., *sig]"""
    lead = coef.shape[:-(nd + 1)]
    out = _Synthesis.apply(coef.reshape(-1, *coef.shape[-(nd + 1):]), wave, mode, nd)
    return out.reshape(*lead, *out.shape[1:])


def _crop_like(ll, ref_shape, nd):
    """An approximation band one sample longer than the next-finer detail band (odd length one level up) drops its
    last sample before the synthesis (pytorch_wavelets DWTInverse.forward; ptwt waverec3 padding removal)."""
    for ax in range(-nd, 0):
        if ll.shape[ax] == ref_shape[ax] + 1:
            ll = ll.narrow(ax, 0, ref_shape[ax])
    return ll


# ----------------------------------------------------------------------------------------------------- pytorch_wavelets look-alikes
class DWTForward(nn.Module):
    """pytorch_wavelets.DWTForward(J, wave, mode): x [N, C, H, W] -> (Yl, [Yh_1 (finest) ... Yh_J]) with
    Yh_j [N, C, 3, H_j, W_j]; every level re-analyses the LL band of the previous one (burgers/wave_trans.py:94-98)."""

    def __init__(self, J=1, wave='db1', mode='zero'):
        super().__init__()
        if int(J) < 1:
            raise ValueError('J must be >= 1')
        self.J, self.wave, self.mode = int(J), _wave_name(wave), mode

    def forward(self, x):
        yh = []
        for _ in range(self.J):
            packed = dwt_packed(x, self.wave, self.mode, 2)     # [N, C, 4, H', W']
            x = packed[:, :, 0]
            yh.append(packed[:, :, 1:])
        return x, yh

    def packed(self, x):
        """The same transform returned as the coef_to_tensor tensor [N, C, 4, H', W'] (no stacking copy)."""
        return dwt_packed(x, self.wave, self.mode, 2)


class DWTInverse(nn.Module):
    """pytorch_wavelets.DWTInverse(wave, mode): (Yl, [Yh]) -> x [N, C, 2H' (or 2H'-L+2), ...]."""

    def __init__(self, wave='db1', mode='zero'):
        super().__init__()
        self.wave, self.mode = _wave_name(wave), mode

    def forward(self, coeffs):
        yl, yh = coeffs
        for h in yh[::-1]:
            if h is None:
                h = torch.zeros(yl.shape[0], yl.shape[1], 3, yl.shape[-2], yl.shape[-1], device=yl.device, dtype=yl.dtype)
            yl = idwt_packed(torch.cat([_crop_like(yl, h.shape, 2).unsqueeze(2), h], dim=2), self.wave, self.mode, 2)
        return yl


class DWT1DForward(nn.Module):
    """pytorch_wavelets.DWT1DForward(J): x [N, C, L] -> (lo, [hi_1 (finest) ... hi_J])"""

    def __init__(self, J=1, wave='db1', mode='zero'):
        super().__init__()
        if int(J) < 1:
            raise ValueError('J must be >= 1')
        self.J, self.wave, self.mode = int(J), _wave_name(wave), mode

    def forward(self, x):
        highs = []
        for _ in range(self.J):
            packed = dwt_packed(x, self.wave, self.mode, 1)     # [N, C, 2, L']
            x = packed[:, :, 0]
            highs.append(packed[:, :, 1])
        return x, highs


class DWT1DInverse(nn.Module):
    def __init__(self, wave='db1', mode='zero'):
        super().__init__()
        self.wave, self.mode = _wave_name(wave), mode

    def forward(self, coeffs):
        lo, highs = coeffs
        for hi in highs[::-1]:
            if hi is None:
                hi = torch.zeros_like(lo)
            lo = idwt_packed(torch.stack([_crop_like(lo, hi.shape, 1), hi], dim=2), self.wave, self.mode, 1)
        return lo


# ----------------------------------------------------------------------------------------------------- ptwt look-alikes
def wavedec3(data, wavelet, mode='zero', level=1):
    """ptwt.wavedec3: [N, T, H, W] -> [lll, {coarsest 'aad'..'ddd'}, ..., {finest}] (dicts in BANDS3 order)."""
    if level is None or int(level) < 1:
        raise ValueError('level must be >= 1')
    dicts = []
    for _ in range(int(level)):
        packed = dwt_packed(data, _wave_name(wavelet), mode, 3)     # [N, 8, T', H', W']
        data = packed[:, 0]
        dicts.append({k: packed[:, i + 1] for i, k in enumerate(BANDS3[1:])})
    return [data] + dicts[::-1]


def wavedec3_packed(data, wavelet, mode='zero'):
    """The smoke coef_to_tensor tensor [N, 8, T', H', W'] directly."""
    return dwt_packed(data, _wave_name(wavelet), mode, 3)


def waverec3(coeffs, wavelet, mode='zero'):
    """ptwt.waverec3: coefficient list [lll, {coarsest}, ..., {finest}] -> [N, T, H, W]."""
    lll = coeffs[0]
    for det in coeffs[1:]:
        packed = torch.stack([_crop_like(lll, det['aad'].shape, 3)] + [det[k] for k in BANDS3[1:]], dim=1)
        lll = idwt_packed(packed, _wave_name(wavelet), mode, 3)
    return lll
