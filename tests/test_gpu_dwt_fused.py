"""The fused single-launch DWT / IDWT kernels (csrc/dwt.hip, dwt_*_fused_kernel) against the per-axis kernels they replace.

Both run the same pass order and the same fmaf chains, so the comparison is BIT-EXACT (torch.equal) for all four entry points
(fwd, inv, inv_adjoint, fwd_adjoint), every embedded wavelet (L = 2 .. 10), both boundary modes, odd and ragged sizes, and padded
(strided) coefficient destinations. The per-axis path itself is pinned against PyWavelets in tests/test_gpu_ops.py, which now runs
through the fused kernels by default. GPU box only."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def W():
    from wdno_amd import wavelets, _lib
    return wavelets, _lib.load()


def both(lib, fn):
    """fn() through the fused kernels (default) and through the per-axis passes (debug switch 11)."""
    a = fn()
    lib.wdno_set_debug(11)
    try:
        b = fn()
    finally:
        lib.wdno_set_debug(0)
    torch.cuda.synchronize()
    return a, b


CASES = [
    # nd, mode, wave, signal shape (leading dims, then nd transformed dims)
    (2, 'periodization', 'bior2.4', (3, 2, 81, 120)),       # the reference's Burgers size (odd rows: repeated last sample)
    (2, 'periodization', 'bior2.4', (4, 2, 160, 128)),      # BASELINE synthetic size
    (2, 'periodization', 'db4', (2, 40, 36)),
    (2, 'periodization', 'haar', (2, 10, 12)),
    (2, 'periodization', 'bior2.2', (5, 33, 18)),
    (2, 'periodization', 'sym4', (1, 3, 24, 50)),
    (2, 'zero', 'bior1.3', (3, 1, 64, 64)),                 # initial-density transform (W6)
    (2, 'zero', 'bior2.4', (2, 37, 21)),
    (2, 'zero', 'db2', (2, 9, 11)),
    (2, 'zero', 'haar', (2, 8, 6)),
    (3, 'zero', 'bior1.3', (5, 32, 64, 64)),                # one smoke simulation
    (3, 'zero', 'bior1.3', (2, 9, 11, 13)),
    (3, 'zero', 'bior1.3', (2, 64, 128, 128)),              # super-resolution size (config 5)
    (3, 'zero', 'db4', (2, 12, 20, 18)),
    (3, 'zero', 'haar', (3, 6, 8, 10)),
    (3, 'periodization', 'bior2.4', (2, 16, 20, 24)),
    (3, 'periodization', 'bior2.2', (2, 7, 9, 12)),
    (3, 'periodization', 'db3', (1, 8, 10, 6)),
]


@pytest.mark.parametrize('nd,mode,wave,shape', CASES)
def test_fused_equals_per_axis_bit_exact(W, nd, mode, wave, shape):
    wavelets, lib = W
    g = torch.Generator().manual_seed(hash((nd, mode, wave, shape)) % 1000)
    x = torch.randn(*shape, generator=g).to(DEV)
    fa, fb = both(lib, lambda: wavelets.dwt_packed(x, wave, mode, nd))
    assert torch.equal(fa, fb), 'forward'
    coef = torch.randn(*fa.shape, generator=g).to(DEV)
    ia, ib = both(lib, lambda: wavelets.idwt_packed(coef, wave, mode, nd))
    assert torch.equal(ia, ib), 'inverse'
    # adjoints through autograd
    def grads():
        xx = x.clone().requires_grad_(True)
        cc = coef.clone().requires_grad_(True)
        y = wavelets.dwt_packed(xx, wave, mode, nd)
        z = wavelets.idwt_packed(cc, wave, mode, nd)
        gy = torch.cos(torch.arange(y.numel(), device=DEV, dtype=torch.float32)).reshape(y.shape)
        gz = torch.sin(torch.arange(z.numel(), device=DEV, dtype=torch.float32)).reshape(z.shape)
        (y * gy).sum().backward()
        (z * gz).sum().backward()
        return xx.grad, cc.grad
    (ga, gca), (gb, gcb) = both(lib, grads)
    assert torch.equal(ga, gb), 'forward adjoint'
    assert torch.equal(gca, gcb), 'inverse adjoint'


def test_fused_padded_destination_and_untouched_padding(W):
    """The packed store may go into a padded tensor (strides in wdno_dwt_desc): padding must stay untouched."""
    wavelets, lib = W
    from wdno_amd import ops
    from wdno_amd.wavelets import _filters
    x = torch.randn(3, 32, 64, 64, device=DEV)
    filt, L = _filters('bior1.3')
    dst = torch.full((3, 8, 24, 40, 40), 7.0, device=DEV)
    cs = (8 * 24 * 40 * 40, 24 * 40 * 40, 40 * 40, 40)
    ops.dwt_call('fwd', x, dst, 3, 1, filt, 3, [32, 64, 64], [18, 34, 34], cs)
    ref = wavelets.dwt_packed(x, 'bior1.3', 'zero', 3)
    assert torch.equal(dst[:, :, :18, :34, :34], ref)
    mask = torch.ones_like(dst, dtype=torch.bool)
    mask[:, :, :18, :34, :34] = False
    assert bool((dst[mask] == 7.0).all())
    back = torch.empty(3, 32, 64, 64, device=DEV)
    ops.dwt_call('inv', dst, back, 3, 1, filt, 3, [32, 64, 64], [18, 34, 34], cs)
    assert (back - x).abs().max() < 1e-5


def test_roundtrip_full_size_properties(W):
    """BASELINE sizes: perfect reconstruction and linearity (size-independent properties)."""
    wavelets, lib = W
    x = torch.randn(64, 2, 160, 128, device=DEV)
    y = torch.randn(64, 2, 160, 128, device=DEV)
    c = wavelets.dwt_packed(x, 'bior2.4', 'periodization', 2)
    assert c.shape == (64, 2, 4, 80, 64)
    assert (wavelets.idwt_packed(c, 'bior2.4', 'periodization', 2) - x).abs().max() < 2e-5
    cl = wavelets.dwt_packed(2.0 * x + y, 'bior2.4', 'periodization', 2)
    assert (cl - (2.0 * c + wavelets.dwt_packed(y, 'bior2.4', 'periodization', 2))).abs().max() < 3e-5
    x3 = torch.randn(32, 32, 64, 64, device=DEV)
    c3 = wavelets.dwt_packed(x3, 'bior1.3', 'zero', 3)
    assert c3.shape == (32, 8, 18, 34, 34)
    assert (wavelets.idwt_packed(c3, 'bior1.3', 'zero', 3) - x3).abs().max() < 2e-5


@pytest.mark.parametrize('mode,wave,shape', [c[1:] for c in CASES if c[0] == 3])
def test_streaming_3d_synthesis_equals_the_all_frames_kernel(W, mode, wave, shape):
    """Round 3: the 3-D synthesis streams over the coefficient frames of its tile (dwt_synthesis3_stream_kernel: one frame in LDS, the T
    pass accumulated in registers in descending frame order) -- same terms in the same order as the round-2 kernel that kept all frames
    in LDS (debug switch 45) and as the per-axis passes: bit-exact, including the synthesis-shaped forward adjoint."""
    wavelets, lib = W
    g = torch.Generator().manual_seed(len(wave) + shape[-1])
    x = torch.randn(*shape, generator=g).to(DEV)
    coef = torch.randn_like(wavelets.dwt_packed(x, wave, mode, 3))

    def run():
        cc = coef.clone().requires_grad_(True)
        xx = x.clone().requires_grad_(True)
        z = wavelets.idwt_packed(cc, wave, mode, 3)
        y = wavelets.dwt_packed(xx, wave, mode, 3)
        (y * torch.cos(torch.arange(y.numel(), device=DEV, dtype=torch.float32)).reshape(y.shape)).sum().backward()
        return z.detach(), xx.grad
    a = run()
    lib.wdno_set_debug(45)
    try:
        b = run()
    finally:
        lib.wdno_set_debug(0)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
