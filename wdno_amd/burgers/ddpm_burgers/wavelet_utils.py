"""burgers/ddpm_burgers/wavelet_utils.py:5-28 on the GPU."""
from wdno_amd import ops


def upsample_coef(w_sub, shape):
    """Nearest x2 of wavelet coefficients along (t, x): [N, l, nt, nx] -> [N, l, 2 nt, 2 nx] (one HIP gather launch)."""
    n, l, nt, nx = w_sub.shape
    return ops.upsample_coef_raw(w_sub, n * l, nt, 1, 1, nx, 2, 1, 2).reshape(n, l, 2 * nt, 2 * nx)


def get_wt_T(test_data, shape):
    """Coefficients of the last physical time row per level (pure indexing)."""
    wt_T = [test_data[:, 0, shape[-1][-2] - 1]]
    for i in range(len(shape)):
        wt_T.append(test_data[:, 1 + 3 * i:1 + 3 * (i + 1), shape[i][-2] - 1])
    return wt_T
