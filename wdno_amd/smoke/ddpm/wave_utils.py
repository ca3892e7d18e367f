"""smoke/ddpm/wave_utils.py:1-14 on the GPU."""
from wdno_amd import ops


def upsample_coef(w_sub, shape, type):
    """[N, nt, l, nx, nx] -> x2 in time ('time') or in both space axes (anything else); one HIP gather launch."""
    n, nt, l, nx = w_sub.shape[0], w_sub.shape[1], w_sub.shape[2], w_sub.shape[3]
    ny = w_sub.shape[4]
    if type == 'time':
        return ops.upsample_coef_raw(w_sub, n, nt, l * nx * ny, 1, 1, 2, 1, 1).reshape(n, 2 * nt, l, nx, ny)
    return ops.upsample_coef_raw(w_sub, n * nt * l, nx, 1, 1, ny, 2, 1, 2).reshape(n, nt, l, 2 * nx, 2 * ny)
