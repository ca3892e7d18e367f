"""Parity of every C-ABI operator (through wdno_amd.ops) against a plain PyTorch / oracle CPU reference.

GPU box only (`-m gpu`). fp32 kernels are compared with float64 CPU references; tolerances are rel-L2 <= 2e-6 for
single operators (the north-star bar is 1e-5 end to end), bit-exact for pure layout / index work.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import load_npz, rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TOL = 2e-6


@pytest.fixture(scope='module')
def ops():
    from wdno_amd import ops as o
    o._lib_()          # fails loudly if libwdno_hip.so is missing
    return o


def g(shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64) * scale)


def dev(t, grad=False):
    x = t.to(torch.float32).to(DEV).contiguous()
    return x.requires_grad_(True) if grad else x


def to_cl(x):      # [N, C, *sp] -> [N, *sp, C]
    return x.permute(0, *range(2, x.dim()), 1).contiguous()


def from_cl(x):
    return x.permute(0, x.dim() - 1, *range(1, x.dim() - 1)).contiguous()


# ----------------------------------------------------------------------------------------------------- layout (bit exact)
@pytest.mark.parametrize('shape,cp', [((3, 42, 5, 7), 44), ((2, 9, 16, 16), 12), ((1, 64, 33), 64), ((4, 3, 8, 12), 4), ((5, 42, 40, 40), 44), ((2, 82, 9, 13), 96),
                                      ((2, 128, 70), 128), ((1, 200, 65), 256), ((3, 1, 130), 1)])
def test_layout_roundtrip(ops, shape, cp):
    x = g(shape, 1).float()
    y = ops.nc_to_cl(x.to(DEV), cp)
    ref = torch.zeros(shape[0], *shape[2:], cp)
    ref[..., :shape[1]] = to_cl(x)
    assert torch.equal(y.cpu(), ref)
    z = ops.cl_to_nc(y, shape[1])
    assert torch.equal(z.cpu(), x)


def test_concat_upsample_exact(ops):
    a, b = g((2, 5, 6, 8), 2).float(), g((2, 5, 6, 12), 3).float()
    c = ops.concat_cl(a.to(DEV), b.to(DEV))
    assert torch.equal(c.cpu(), torch.cat([a, b], dim=-1))
    x = dev(g((2, 3, 5, 8), 4), grad=True)
    y = ops.upsample2x_cl(x)
    ref = F.interpolate(from_cl(x.detach().cpu()), scale_factor=2, mode='nearest')
    assert torch.equal(from_cl(y.detach().cpu()), ref)
    go = g(tuple(y.shape), 5).float()
    y.backward(go.to(DEV))
    xr = from_cl(x.detach().cpu()).requires_grad_(True)
    F.interpolate(xr, scale_factor=2, mode='nearest').backward(from_cl(go))
    assert rel_l2(from_cl(x.grad.cpu()), xr.grad) < 1e-6


# ----------------------------------------------------------------------------------------------------- pointwise
def test_activations_and_add(ops):
    x = g((1000,), 6, 3.0)
    for fn, ref in ((ops.silu, F.silu), (ops.gelu, F.gelu)):
        xd = dev(x, grad=True)
        y = fn(xd)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        assert rel_l2(y.detach(), yr.detach()) < TOL
        go = g((1000,), 7)
        y.backward(dev(go)); yr.backward(go)
        assert rel_l2(xd.grad, xr.grad) < TOL
    a, b = g((77,), 8), g((77,), 9)
    assert torch.equal(ops.add(dev(a), dev(b)).cpu(), a.float() + b.float())


def test_sinusoidal(ops):
    from oracle.unet_ref import sinusoidal_embedding
    t = torch.tensor([0, 1, 17, 500, 999])
    for dim in (8, 64, 128):
        out = ops.sinusoidal_embedding(t.to(DEV), dim)
        ref = sinusoidal_embedding(t, dim)
        assert (out.cpu() - ref).abs().max() < 2e-6


# ----------------------------------------------------------------------------------------------------- convolution
def conv_case(ops, xs, ws, stride, padding, seed, transposed=False, bias=True, residual=False, tol=TOL):
    nd = len(ws) - 2
    x = g(xs, seed)
    w = g(ws, seed + 1, 1.0 / math.sqrt(np.prod(ws[1:])))
    b = g((ws[1] if transposed else ws[0],), seed + 2) if bias else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    if transposed:
        yr = F.conv_transpose3d(xr, wr, br, stride=stride, padding=padding)
    elif nd == 3:
        yr = F.conv3d(xr, wr, br, stride=stride, padding=padding)
    elif nd == 2:
        yr = F.conv2d(xr, wr, br, stride=stride, padding=padding)
    else:
        yr = F.linear(xr, wr, br)
    res = g(tuple(yr.shape), seed + 3) if residual else None
    if residual:
        rr = res.clone().requires_grad_(True)
        yr = yr + rr
    go = g(tuple(yr.shape), seed + 4)
    yr.backward(go)

    cpad = ops.pad4(xs[1]) if nd else ops.pad4(xs[-1])
    if nd:
        xd = torch.zeros(xs[0], *xs[2:], cpad, dtype=torch.float64)
        xd[..., :xs[1]] = to_cl(x)
    else:
        xd = torch.zeros(*xs[:-1], cpad, dtype=torch.float64)
        xd[..., :xs[-1]] = x
    xd = dev(xd, grad=True)
    wd, bd = dev(w, grad=True), (dev(b, grad=True) if bias else None)
    k = yr.shape[1] if nd else yr.shape[-1]
    kp = ops.pad4(k)
    rd = None
    if residual:
        rd = torch.zeros(*((yr.shape[0], *yr.shape[2:], kp) if nd else (*yr.shape[:-1], kp)), dtype=torch.float64)
        rd[..., :k] = to_cl(res) if nd else res
        rd = dev(rd, grad=True)
    if transposed:
        y = ops.conv_transpose_cl(xd, wd, bd)
    else:
        y = ops.conv_cl(xd, wd, bd, stride=stride, padding=padding, residual=rd)
    yc = from_cl(y.detach().cpu())[:, :k] if nd else y.detach().cpu()[..., :k]
    assert rel_l2(yc, yr.detach()) < tol, 'forward'
    god = torch.zeros(tuple(y.shape), dtype=torch.float64)
    god[..., :k] = to_cl(go) if nd else go
    y.backward(dev(god))
    gx = from_cl(xd.grad.cpu())[:, :xs[1]] if nd else xd.grad.cpu()[..., :xs[-1]]
    assert rel_l2(gx, xr.grad) < tol, 'dgrad'
    assert rel_l2(wd.grad, wr.grad) < tol, 'wgrad'
    if bias:
        assert rel_l2(bd.grad, br.grad) < tol, 'bias grad'
    if residual:
        assert rel_l2((from_cl(rd.grad.cpu())[:, :k] if nd else rd.grad.cpu()[..., :k]), rr.grad) < tol, 'residual grad'


CONV_CASES = [
    # (x shape NC..., weight shape, stride, padding)
    ('3x3x3_64', (1, 64, 4, 10, 12), (64, 64, 3, 3, 3), 1, 1),
    ('3x3x3_wide', (2, 128, 3, 6, 6), (256, 128, 3, 3, 3), 1, 1),
    ('3x3x3_tiny', (2, 8, 4, 8, 8), (16, 8, 3, 3, 3), 1, 1),
    ('3x3x3_oddC', (1, 36, 3, 5, 7), (20, 36, 3, 3, 3), 1, 1),
    ('7x7x7_init', (1, 42, 6, 9, 9), (64, 42, 7, 7, 7), 1, 3),
    ('1x1x1', (2, 64, 3, 7, 5), (42, 64, 1, 1, 1), 1, 0),
    ('down_144', (2, 16, 3, 8, 12), (16, 16, 1, 4, 4), (1, 2, 2), (0, 1, 1)),
    ('down_144_64', (1, 64, 2, 20, 20), (64, 64, 1, 4, 4), (1, 2, 2), (0, 1, 1)),
    ('2d_3x3', (2, 128, 16, 12), (128, 128, 3, 3), 1, 1),
    ('2d_3x3_cat', (1, 384, 8, 8), (256, 384, 3, 3), 1, 1),
    ('2d_7x7', (2, 9, 16, 16), (128, 9, 7, 7), 1, 3),
    ('2d_1x1', (2, 128, 8, 8), (384, 128, 1, 1), 1, 0),
    ('2d_patch2', (2, 8, 8, 12), (16, 8, 2, 2), 2, 0),
]


# large enough (>= 1024 output pixels, reduction >= 128) to take the 3 x fp16-split MFMA path in forward and dgrad
CONV_CASES_H3 = [
    ('h3_3x3x3_64', (1, 64, 8, 20, 20), (64, 64, 3, 3, 3), 1, 1),
    ('h3_3x3x3_wide', (2, 128, 4, 16, 16), (256, 128, 3, 3, 3), 1, 1),
    ('h3_3x3x3_oddC', (1, 36, 8, 16, 16), (20, 36, 3, 3, 3), 1, 1),
    ('h3_7x7x7_init', (1, 42, 8, 16, 16), (64, 42, 7, 7, 7), 1, 3),
    ('h3_1x1x1', (4, 256, 4, 16, 16), (128, 256, 1, 1, 1), 1, 0),
    ('h3_2d_3x3', (8, 128, 16, 16), (128, 128, 3, 3), 1, 1),
    ('h3_2d_7x7', (8, 9, 16, 16), (128, 9, 7, 7), 1, 3),
    ('h3_down_144', (2, 64, 4, 32, 32), (64, 64, 1, 4, 4), (1, 2, 2), (0, 1, 1)),
]


# more tiles / work items than CUs: every persistent block of the LDS-DMA kernels walks over several tiles (forward, dgrad)
# and several (tap, split) items (wgrad), with the producers prefetching across the tile boundary
CONV_CASES_PERSISTENT = [
    ('persist_k32', (2, 16, 24, 40, 40), (32, 16, 3, 3, 3), 1, 1),        # 300 tiles of 256 x 64 -> run as 400 tiles of 192 x 64
    ('persist_k32b', (4, 16, 24, 40, 40), (32, 16, 3, 3, 3), 1, 1),       # 600 tiles of 256 x 64 (three rounds; 192-row tiles would need four)
    ('persist_k96', (1, 16, 24, 40, 40), (96, 16, 3, 3, 3), 1, 1),        # 300 tiles of 128 x 128 -> run as 200 tiles of 192 x 128
    ('persist_k96b', (2, 16, 24, 40, 40), (96, 16, 3, 3, 3), 1, 1),       # 600 tiles of 128 x 128 (three rounds either way)
    ('persist_1x1', (2, 64, 24, 40, 40), (24, 64, 1, 1, 1), 1, 0),        # one step per tile
]


@pytest.mark.parametrize('name,xs,ws,stride,padding', CONV_CASES_PERSISTENT, ids=[c[0] for c in CONV_CASES_PERSISTENT])
def test_conv_f16x3_persistent_blocks(ops, name, xs, ws, stride, padding):
    conv_case(ops, xs, ws, stride, padding, seed=sum(name.encode()) % 1000)


@pytest.mark.parametrize('name,xs,ws,stride,padding', CONV_CASES_H3, ids=[c[0] for c in CONV_CASES_H3])
def test_conv_f16x3_path(ops, name, xs, ws, stride, padding):
    assert ops.CONV_MATH == 'f16x3'
    n_launch = {}
    ops.PROFILE = n_launch
    try:
        conv_case(ops, xs, ws, stride, padding, seed=sum(name.encode()) % 1000)
    finally:
        ops.PROFILE = None
    assert any('h3' in k for k in n_launch), f'fp16-split kernel was not used: {list(n_launch)}'


@pytest.mark.parametrize('mode', [7], ids=['dma'])
@pytest.mark.parametrize('name,xs,ws,stride,padding', CONV_CASES_H3, ids=[c[0] for c in CONV_CASES_H3])
def test_conv_f16x3_dma_kernels(ops, name, xs, ws, stride, padding, mode):
    """The LDS-DMA forward / dgrad kernels (normally picked for >= 256 tiles) forced on the small parity cases:
    ragged M / N tiles, C % 32 != 0 (per-lane dx), 7-wide taps, strided and 1x1 geometries; several tiles per persistent block."""
    lib = ops._lib_()
    lib.wdno_set_debug(mode)
    try:
        conv_case(ops, xs, ws, stride, padding, seed=sum(name.encode()) % 1000)
    finally:
        lib.wdno_set_debug(0)


# stride-1 convolutions on equal grids with whole 32-channel blocks: the tap-resident kernels (csrc/conv_h3t.hip)
CONV_CASES_TAP = [
    ('tap_3d_64', (2, 64, 5, 9, 7), (64, 64, 3, 3, 3), 1, 1),              # rows of 7 pixels: every tile crosses rows, planes and the batch
    ('tap_3d_c96_k40', (1, 96, 4, 10, 10), (40, 96, 3, 3, 3), 1, 1),       # three channel blocks, ragged output channels
    ('tap_2d_c32', (3, 32, 17, 13), (72, 32, 3, 3), 1, 1),                 # one channel block, two N tiles (ragged)
    ('tap_3d_c128_k40', (1, 128, 4, 10, 10), (40, 128, 3, 3, 3), 1, 1),    # two 64-channel tiles in the weight gradient, ragged k
    ('tap_2d_c64_w5', (5, 64, 9, 5), (64, 64, 3, 3), 1, 1),                # 5-pixel rows
    ('tap_3d_133', (1, 64, 6, 12, 12), (128, 64, 1, 3, 3), 1, (0, 1, 1)),
    ('tap_3d_313', (1, 64, 6, 12, 12), (64, 64, 3, 1, 3), 1, (1, 0, 1)),
    ('tap_3d_w2', (1, 32, 3, 4, 2), (32, 32, 3, 3, 3), 1, 1),              # both W-neighbours missing somewhere in every row
    ('tap_3d_multi', (8, 64, 6, 40, 40), (64, 64, 3, 3, 3), 1, 1),         # 300 tiles: persistent blocks prefetch across tile boundaries
    ('tap_3d_multi_wide', (4, 128, 6, 20, 20), (192, 128, 3, 3, 3), 1, 1),
    # round 6, split-pair items of the window weight gradient (the odd tap row paired across two pixel splits; 'tap_3d_multi' above takes them too):
    ('tap_3d_splitpair_odd', (3, 64, 6, 40, 40), (64, 64, 3, 3, 3), 1, 1),  # 53 splits: the last pair holds one window
    ('tap_2d_splitpair', (16, 128, 64, 64), (128, 128, 3, 3), 1, 1),        # three tap rows = one ordinary pair + split pairs; two k tiles x two channel tiles
    ('tap_3d_splitpair_k40', (2, 128, 6, 40, 40), (40, 128, 3, 3, 3), 1, 1),  # ragged output channels, two channel tiles
    # 7-wide taps on 48 plane channels (the smoke stem): 16-channel-block forward kernel, seven-wide window weight gradient
    ('tap_stem_w7', (2, 42, 5, 9, 7), (64, 42, 7, 7, 7), 1, 3),            # rows shorter than the 32-pixel step
    ('tap_stem_k40', (1, 42, 4, 10, 33), (40, 42, 7, 7, 7), 1, 3),         # ragged output channels, odd row length
    ('tap_stem_2d', (3, 48, 20, 36), (64, 48, 7, 7), 1, 3),
    ('tap_stem_multi', (2, 42, 6, 40, 40), (64, 42, 7, 7, 7), 1, 3),       # 19200 pixels
    ('tap_stem_c82', (1, 82, 4, 18, 18), (64, 82, 7, 7, 7), 1, 3),         # the super-resolution stem: 82 channels in planes of 96
]


@pytest.mark.parametrize('mode', [7, 8], ids=['tap', 'chunked'])
@pytest.mark.parametrize('name,xs,ws,stride,padding', CONV_CASES_TAP, ids=[c[0] for c in CONV_CASES_TAP])
def test_conv_f16x3_tap_resident_kernels(ops, name, xs, ws, stride, padding, mode):
    """debug 7 forces the LDS-DMA kernels (these geometries then take the tap-resident one); debug 8 keeps them on the chunked
    kernel of conv_h3d.hip: both must meet the fp32 tolerance against the float64 reference."""
    lib = ops._lib_()
    lib.wdno_set_debug(mode)
    try:
        conv_case(ops, xs, ws, stride, padding, seed=sum(name.encode()) % 1000)
    finally:
        lib.wdno_set_debug(0)


@pytest.mark.parametrize('mode', [30, 31], ids=['160x128', '320x64'])
@pytest.mark.parametrize('name,xs,ws,stride,padding', [c for c in CONV_CASES_TAP if c[2][-1] == 3], ids=[c[0] for c in CONV_CASES_TAP if c[2][-1] == 3])
def test_conv_f16x3_tap_five_row_tiles(ops, name, xs, ws, stride, padding, mode):
    """The 160 x 128 (one wave row x four wave columns) and 320 x 64 shapes of the tap-resident kernel, which the launch plan picks for
    layers whose tile count leaves the last round of the persistent grid mostly empty; debug 30 / 31 force them on the small cases."""
    lib = ops._lib_()
    lib.wdno_set_debug(mode)
    try:
        conv_case(ops, xs, ws, stride, padding, seed=sum(name.encode()) % 1000)
    finally:
        lib.wdno_set_debug(0)


# few pixels x many channels with a long reduction (the deep levels of the Burgers U-Net): 128 x 128 tiles, reduction cut into runs of stages
CONV_CASES_SPLIT = [
    ('split_2d_512', (16, 512, 8, 8), (512, 512, 3, 3), 1, 1, '/4'),       # 32 tiles x 4 runs of 12 stages
    ('split_2d_ragged', (17, 512, 8, 8), (520, 512, 3, 3), 1, 1, '/4'),    # 1088 pixels (half a tile at the end), ragged last column of tiles
    ('split_3d_256', (4, 256, 4, 8, 8), (256, 256, 3, 3, 3), 1, 1, '/4'),  # 72 stages in 4 runs of 18
]


@pytest.mark.parametrize('residual', [False, True], ids=['plain', 'residual'])
@pytest.mark.parametrize('name,xs,ws,stride,padding,tag', CONV_CASES_SPLIT, ids=[c[0] for c in CONV_CASES_SPLIT])
def test_conv_f16x3_split_reduction(ops, name, xs, ws, stride, padding, tag, residual):
    """csrc/conv_h3t.hip with p.tsplit > 1: the runs' partial sums go through the caller's workspace and are added in a fixed order (forward
    and data gradient both pass here); same tolerance as every other kernel, the split must have been taken, and two runs agree bit for bit."""
    n_launch = {}
    ops.PROFILE = n_launch
    try:
        conv_case(ops, xs, ws, stride, padding, seed=sum(name.encode()) % 1000, residual=residual)
    finally:
        ops.PROFILE = None
    assert any(k.endswith(tag) for k in n_launch), f'split reduction was not used: {list(n_launch)}'
    x = dev(to_cl(g(xs, 5)).float().contiguous())
    w = dev(g(ws, 6, 0.02).float())
    y0 = ops.conv_cl(x, w, None, padding=padding)
    y1 = ops.conv_cl(x, w, None, padding=padding)
    assert torch.equal(y0, y1)
    lib = ops._lib_()
    lib.wdno_set_debug(56)                           # the unsplit kernels on the same operands
    try:
        y2 = ops.conv_cl(x, w, None, padding=padding)
    finally:
        lib.wdno_set_debug(0)
    assert rel_l2(y0.double().cpu(), y2.double().cpu()) < TOL       # two fp32 accumulation orders of a 4608-term sum


@pytest.mark.parametrize('xs,ws', [((4, 64, 32, 32), (128, 64, 2, 2)), ((6, 40, 1, 36, 20), (72, 40, 1, 2, 2))], ids=['2d', '3d_ragged'])
def test_conv_patch2_dgrad_on_split_kernels(ops, xs, ws):
    """Data gradient of the folded (1,2,2) / stride-2 Downsample (Burgers unet.py:64-68) as four 1x1 convolutions of the planes of dy, the
    operand of tap (py, px) gathered from the weight by the pack kernel (kind 'q{py}{px}'): fp32 tolerance against float64 autograd, the
    split kernels must have run, and the exact-fp32 path (the knob off) agrees."""
    nd = len(ws) - 2
    stride = 2 if nd == 2 else (1, 2, 2)
    n_launch = {}
    ops.PROFILE = n_launch
    try:
        conv_case(ops, xs, ws, stride, 0, seed=77)
    finally:
        ops.PROFILE = None
    assert sum(len(v) for k, v in n_launch.items() if 'h3' in k) >= 5, list(n_launch)      # forward + four taps
    x = dev(to_cl(g(xs, 5)), grad=True)
    w = dev(g(ws, 6, 0.05))
    go = dev(to_cl(g((xs[0], ws[0], *[v // 2 if i >= len(xs) - 4 else v for i, v in enumerate(xs[2:])]), 7)))
    grads = []
    for knob in (True, False):
        ops.PATCH_DGRAD_H3 = knob
        try:
            x.grad = None
            ops.conv_cl(x, w, None, stride=stride, padding=0).backward(go)
            grads.append(x.grad.clone())
        finally:
            ops.PATCH_DGRAD_H3 = True
    assert rel_l2(grads[0].double().cpu(), grads[1].double().cpu()) < TOL


def test_conv_f16x3_wide_dynamic_range(ops):
    """Gradient-like magnitudes (1e-7) and large activations (1e3) must survive the per-tensor scaling."""
    for scale in (1e-7, 1.0, 1e3):
        x = g((1, 64, 8, 16, 16), 90) * scale
        w = g((64, 64, 3, 3, 3), 91) * 0.02
        yr = F.conv3d(x, w, None, padding=1)
        xd = dev(to_cl(x))
        y = ops.conv_cl(xd, dev(w), None, padding=1)
        assert rel_l2(from_cl(y.cpu()), yr) < TOL, scale


@pytest.mark.parametrize('k,c', [(128, 64), (72, 40), (256, 256)])
def test_pack_split_weight_patch_taps(ops, k, c):
    """Pack modes 6-9 (kind 'q{py}{px}'): tap (py, px) of a [K, C, 1, 2, 2] weight as the data-gradient operand [Cp][K8] of a 1x1
    convolution -- hi + lo planes reconstruct w[k][c][0][py][px] to 2^-21 of max|w|, padding rows / columns are zero."""
    w = dev(g((k, c, 1, 2, 2), 31, 0.3))
    cp, k8 = ops.pad4(c), ops.pad8(k)
    for py in range(2):
        for px in range(2):
            hi, lo, sc = ops.split_weight(w, f'q{py}{px}', k8, cp)
            assert hi.shape == (cp, k8) and lo.shape == (cp, k8)
            rec = (hi.double() + lo.double()) / float(sc)
            want = torch.zeros(cp, k8, dtype=torch.float64, device=DEV)
            want[:c, :k] = w[:, :, 0, py, px].double().t()
            assert float((rec - want).abs().max()) <= 2.0 ** -21 * float(w.abs().max())
            assert float(rec[c:].abs().max() if cp > c else 0.0) == 0.0 and float(rec[:, k:].abs().max() if k8 > k else 0.0) == 0.0


@pytest.mark.parametrize('c', [128, 256])
def test_pack_split_weight_fragment_order(ops, c):
    """Pack modes 10 / 11 (kinds 'aq' / 'ao'): the forward operands of to_qkv [384][C] / to_out [C][128] in the fragment order the wide attention
    kernels stream (csrc/attn_fused_wide.hip): the same values as the row-major forward operand (mode 0), 16-byte groups permuted to
    [head][q|k|v][k-step pair][step][lane half][row of 32][8] and [wave][tile][head][step][lane half][row][8] -- bit for bit, also through the
    multi-tensor refresh after an in-place weight update."""
    wq = dev(g((384, c), 51, 0.3))
    wo = dev(g((c, 128), 52, 0.3))
    for rep in range(2):
        fh, fl, fs = ops.split_weight(wq, 'f', c, 384)
        ah, al, as_ = ops.split_weight(wq, 'aq', c, 384)
        assert torch.equal(fs, as_)
        for pf, pa in ((fh, ah), (fl, al)):
            want = pf.view(3, 4, 32, c // 32, 2, 2, 8).permute(1, 0, 3, 5, 4, 2, 6).contiguous().view(384, c)
            assert torch.equal(pa, want)
        fh, fl, fs = ops.split_weight(wo, 'f', 128, c)
        ah, al, as_ = ops.split_weight(wo, 'ao', 128, c)
        assert torch.equal(fs, as_)
        for pf, pa in ((fh, ah), (fl, al)):
            want = pf.view(4, c // 128, 32, 4, 2, 2, 8).permute(0, 1, 3, 5, 4, 2, 6).contiguous().view(c, 128)
            assert torch.equal(pa, want)
        with torch.no_grad():                 # an optimiser step: every operand is refreshed by the two multi-tensor launches
            wq.mul_(1.37); wo.add_(0.01)
        ops.bump_weight_epoch()


def test_conv_split_reduction_without_workspace(ops):
    """wdno_conv_fwd_f16x3_ws with no (or too small a) workspace computes the same convolution unsplit; with the workspace the runs'
    partial sums are added in a fixed order -- the three results agree to the accumulation-order tolerance, the split one bit for bit with itself."""
    import ctypes as C
    from wdno_amd import _lib
    x = dev(to_cl(g((16, 512, 8, 8), 41)).float().contiguous())
    w = dev(g((512, 512, 3, 3), 42, 0.02).float())
    planes = ops.split_f16(x.reshape(-1, 512))
    wh, wl, sw = ops.split_weight(w, 'f', 512, 512)
    geom = ops._geom((16, 1, 8, 8), 512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 8, 8))
    lib = ops._lib_()
    need = int(lib.wdno_conv_fwd_split_ws_bytes(C.byref(geom)))
    assert need == 4 * 1024 * 512 * 4
    outs = []
    for ws_bytes in (need, need, 0, need // 2):
        y = torch.empty(16, 1, 8, 8, 512, device=DEV)
        ws = torch.empty(max(ws_bytes, 4) // 4, device=DEV) if ws_bytes else None
        _lib.check(lib.wdno_conv_fwd_f16x3_ws(ops._p(planes[0]), ops._p(planes[1]), ops._p(planes[2]), ops._p(wh), ops._p(wl), ops._p(sw), None, None,
                                              ops._p(y), None, C.byref(geom), ops._p(ws), ws_bytes, ops._stream()), 'conv_fwd_f16x3_ws')
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[2], outs[3])                      # both unsplit
    assert 0 < rel_l2(outs[0].double().cpu(), outs[2].double().cpu()) < TOL


def test_amax_multi_unaligned_spans(ops):
    """wdno_amax_multi over spans of a flat buffer that start at every 4-byte phase of a 16-byte line and have 0 .. 3 trailing elements
    (the weights of a model live at such offsets of the trainer's flat parameter buffer): equals torch's abs().max() exactly."""
    import ctypes as C
    flat = (torch.randn(70000, generator=torch.Generator().manual_seed(3)) * 3).to(DEV)
    spans = [(0, 1), (1, 2), (3, 3), (6, 5), (11, 4), (15, 1000), (1015, 4097), (5112, 7), (5119, 60001)]
    out = torch.zeros(len(spans), device=DEV)
    items = (ops._AmaxItem * len(spans))()
    for i, (o, n) in enumerate(spans):
        items[i] = ops._AmaxItem(flat.data_ptr() + 4 * o, n, out.data_ptr() + 4 * i)
    tab = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(DEV)
    from wdno_amd import _lib
    _lib.check(ops._lib_().wdno_amax_multi(ops._p(tab), len(spans), 256, ops._stream()), 'amax_multi')
    want = torch.stack([flat[o:o + n].abs().max() for o, n in spans])
    assert torch.equal(out, want), (out, want)


def test_split_f16_reconstruction(ops):
    x = g((300, 44), 92) * torch.logspace(-6, 2, 300, dtype=torch.float64)[:, None]
    hi, lo, s = ops.split_f16(dev(x))
    assert hi.shape == (300, 48) and float(hi[:, 44:].abs().max()) == 0
    rec = (hi.double() + lo.double())[:, :44].cpu() / float(s)
    amax = x.abs().max()
    assert ((rec - x.float().double()).abs().max() / amax) < 2 ** -21
    assert 2 ** 14 <= float(s) * float(x.float().abs().max()) < 2 ** 15


@pytest.mark.parametrize('name,xs,ws,stride,padding', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv(ops, name, xs, ws, stride, padding):
    conv_case(ops, xs, ws, stride, padding, seed=sum(name.encode()) % 1000)


def test_conv_residual_nobias(ops):
    conv_case(ops, (1, 16, 2, 6, 6), (24, 16, 1, 1, 1), 1, 0, seed=11, bias=False, residual=True)


def test_linear(ops):
    conv_case(ops, (37, 64), (256, 64), 1, 0, seed=12)
    conv_case(ops, (2, 5, 3, 128), (64, 128), 1, 0, seed=13, bias=False, residual=True)


@pytest.mark.parametrize('rows,cin,cout,bias', [(8, 256, 512, True), (8, 64, 256, True), (16, 256, 128, False), (1, 32, 12, True),
                                               (3, 128, 256, True), (64, 512, 1024, True), (100, 128, 256, True)])
def test_linear_on_a_few_rows(ops, rows, cin, cout, bias):
    """The time-embedding MLPs (8 samples): the one-wave-per-feature kernels on the unpacked weight, forward, data gradient
    (same kernel on the transposed weight), weight and bias gradient."""
    seen = {}
    orig = ops._linear_rows_backward
    def spy(ctx, gy):
        seen['bwd'] = True
        return orig(ctx, gy)
    ops._linear_rows_backward = spy
    try:
        conv_case(ops, (rows, cin), (cout, cin), 1, 0, seed=300 + rows + cout, bias=bias)
    finally:
        ops._linear_rows_backward = orig
    assert bool(seen.get('bwd')) == (rows <= ops.LINEAR_ROWS_MAX), 'few-rows path up to LINEAR_ROWS_MAX rows, the GEMM kernels beyond'


@pytest.mark.parametrize('c', [16, 64])
def test_conv_transpose(ops, c):
    conv_case(ops, (2, c, 3, 5, 6), (c, c, 1, 4, 4), (1, 2, 2), (0, 1, 1), seed=14 + c, transposed=True)


@pytest.mark.parametrize('cin,cout,sp', [(64, 64, (4, 16, 16)), (128, 72, (3, 20, 20))])
def test_conv_transpose_f16x3_path(ops, cin, cout, sp):
    """Large enough for the split-fp16 kernels: 4 parity-class launches with interleaved output placement, the strided
    data gradient and the weight gradient (operand roles swapped) all on the fp32-equivalent MFMA path."""
    n_launch = {}
    ops.PROFILE = n_launch
    try:
        conv_case(ops, (2, cin, *sp), (cin, cout, 1, 4, 4), (1, 2, 2), (0, 1, 1), seed=200 + cin, transposed=True)
    finally:
        ops.PROFILE = None
    assert any('h3' in k for k in n_launch) and not any(k.startswith('conv_fwd_kernel') for k in n_launch), list(n_launch)


def test_conv_transpose_parity_operands_follow_the_weight(ops):
    """The four parity-class operands of a transposed convolution are gathered from the ConvTranspose weight by the pack kernel and
    refreshed with every other operand after an optimiser step: after an in-place weight change (+ weight epoch) the result must
    follow the new weight."""
    x, w = g((2, 64, 4, 16, 16), 410), g((64, 40, 1, 4, 4), 411) * 0.1
    xd, wd = dev(to_cl(x)), dev(w)
    y1 = ops.conv_transpose_cl(xd, wd)
    assert rel_l2(from_cl(y1.cpu())[:, :40], F.conv_transpose3d(x, w, None, stride=(1, 2, 2), padding=(0, 1, 1))) < TOL
    for k in range(2):
        with torch.no_grad():
            wd.mul_(-0.7).add_(0.01 * (k + 1))
        ops.bump_weight_epoch()
        y2 = ops.conv_transpose_cl(xd, wd)
        assert rel_l2(from_cl(y2.cpu())[:, :40], F.conv_transpose3d(x, wd.detach().cpu().to(x.dtype), None, stride=(1, 2, 2), padding=(0, 1, 1))) < TOL


# ----------------------------------------------------------------------------------------------------- amax records
def _amax_chain(ops, hints, poison=False):
    """layer norm -> 1x1 conv -> group norm + SiLU -> 3x3x3 conv (residual) -> concat -> conv -> add -> conv: every producer that
    can leave an amax record on its output, each followed by a split-fp16 convolution that would use it."""
    ops.AMAX_HINTS = hints
    torch.manual_seed(5)
    n, d, h, w, c = 2, 6, 20, 20, 64
    x = dev(torch.randn(n, d, h, w, c) * 3, True)
    gl = dev(torch.rand(1, c, 1, 1, 1) + 0.5, True)
    w1, w2, w3, w4 = (dev(torch.randn(*s) * 0.05, True) for s in ((c, c, 1, 1, 1), (c, c, 3, 3, 3), (c, 2 * c, 1, 3, 3), (c, c, 1, 3, 3)))
    b2 = dev(torch.randn(c) * 0.1, True)
    gam, bet = dev(torch.rand(c) + 0.5, True), dev(torch.randn(c) * 0.1, True)
    t = ops.layernorm_cl(x, gl)
    t = ops.conv_cl(t, w1)
    t = ops.groupnorm_act(t, gam, bet, 8)
    if poison:
        with torch.no_grad():
            t.mul_(4096.0)                 # an in-place change after the record was left: the record must not be used
    t = ops.conv_cl(t, w2, b2, padding=1, residual=x)
    t = ops.concat_cl(t, x)
    t = ops.conv_cl(t, w3, padding=(0, 1, 1))
    t = ops.add(t, x)
    y = ops.conv_cl(t, w4, padding=(0, 1, 1))
    if not poison:
        y.backward(dev(torch.randn(n, d, h, w, c)))
    ops.AMAX_HINTS = True
    return [y.detach()] + ([] if poison else [v.grad for v in (x, gl, w1, w2, w3, w4, b2, gam, bet)])


def test_amax_records_match_the_sweep(ops):
    """The amax a producer kernel leaves behind is the same number the separate sweep finds, so everything downstream
    (scale, fp16 planes, convolution outputs and all gradients) is bit-identical with and without the records."""
    calls = {}
    orig = ops.tensor_amax
    def counting(x):
        calls['n'] = calls.get('n', 0) + 1
        return orig(x)
    ops.tensor_amax = counting
    try:
        with_rec = _amax_chain(ops, True)
        n_with = calls.pop('n', 0)
        without = _amax_chain(ops, False)
        n_without = calls.pop('n', 0)
    finally:
        ops.tensor_amax = orig
    for a, b in zip(with_rec, without):
        assert torch.equal(a, b)
    assert n_with < n_without and n_with <= n_without - 5, (n_with, n_without)      # the forward chain alone drops 5 sweeps


def test_amax_record_is_void_after_an_in_place_change(ops):
    a = _amax_chain(ops, True, poison=True)[0]
    b = _amax_chain(ops, False, poison=True)[0]
    assert torch.isfinite(a).all() and torch.equal(a, b)


# ----------------------------------------------------------------------------------------------------- normalisation
@pytest.mark.parametrize('shape,groups,use_ss,act', [((2, 64, 3, 10, 10), 8, True, True), ((2, 16, 5, 5), 1, True, True),
                                                      ((1, 8, 2, 4, 4), 4, False, True), ((3, 128, 8, 8), 1, False, False),
                                                      ((2, 1024, 4, 4), 1, True, True)])
def test_groupnorm_act(ops, shape, groups, use_ss, act):
    n, c = shape[0], shape[1]
    x, gam, bet = g(shape, 20, 2.0) + 0.7, g((c,), 21) * 0.3 + 1, g((c,), 22) * 0.3
    ss = g((n, 2 * c), 23, 0.5) if use_ss else None
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gam, bet))
    sr = ss.clone().requires_grad_(True) if use_ss else None
    yr = F.group_norm(xr, groups, gr, br, eps=1e-5)
    if use_ss:
        e = sr.reshape(n, 2 * c, *([1] * (len(shape) - 2)))
        yr = yr * (e[:, :c] + 1) + e[:, c:]
    if act:
        yr = F.silu(yr)
    go = g(shape, 24)
    yr.backward(go)
    xd, gd, bd = dev(to_cl(x), True), dev(gam, True), dev(bet, True)
    sd = dev(ss, True) if use_ss else None
    y = ops.groupnorm_act(xd, gd, bd, groups, sd, act=act)
    assert rel_l2(from_cl(y.detach().cpu()), yr.detach()) < TOL
    y.backward(dev(to_cl(go)))
    assert rel_l2(from_cl(xd.grad.cpu()), xr.grad) < 5e-6
    assert rel_l2(gd.grad, gr.grad) < 5e-6 and rel_l2(bd.grad, br.grad) < 5e-6
    if use_ss:
        assert rel_l2(sd.grad, sr.grad) < 5e-6


@pytest.mark.parametrize('shape,groups,use_ss', [((2, 64, 3, 10, 10), 8, True), ((3, 128, 8, 8), 1, False), ((2, 24, 5, 5), 4, False)])
def test_groupnorm_act_add(ops, shape, groups, use_ss):
    """act(GroupNorm(x)) + residual in one apply pass (identity-skip ResnetBlock tail); channel counts the fused kernel does not take
    (24: C / 8 = 3 is not a power of two) fall back to norm + add. Forward and every gradient against torch."""
    n, c = shape[0], shape[1]
    x, gam, bet, res = g(shape, 50, 2.0) + 0.7, g((c,), 51) * 0.3 + 1, g((c,), 52) * 0.3, g(shape, 55)
    ss = g((n, 2 * c), 53, 0.5) if use_ss else None
    xr, gr, br, rr = (t.clone().requires_grad_(True) for t in (x, gam, bet, res))
    sr = ss.clone().requires_grad_(True) if use_ss else None
    yr = F.group_norm(xr, groups, gr, br, eps=1e-5)
    if use_ss:
        e = sr.reshape(n, 2 * c, *([1] * (len(shape) - 2)))
        yr = yr * (e[:, :c] + 1) + e[:, c:]
    yr = F.silu(yr) + rr
    go = g(shape, 54)
    yr.backward(go)
    xd, gd, bd, rd = dev(to_cl(x), True), dev(gam, True), dev(bet, True), dev(to_cl(res), True)
    sd = dev(ss, True) if use_ss else None
    y = ops.groupnorm_act_add(xd, gd, bd, groups, rd, sd, act=True)
    assert rel_l2(from_cl(y.detach().cpu()), yr.detach()) < TOL
    y.backward(dev(to_cl(go)))
    assert rel_l2(from_cl(xd.grad.cpu()), xr.grad) < 5e-6 and rel_l2(from_cl(rd.grad.cpu()), rr.grad) < 1e-7
    assert rel_l2(gd.grad, gr.grad) < 5e-6 and rel_l2(bd.grad, br.grad) < 5e-6
    if use_ss:
        assert rel_l2(sd.grad, sr.grad) < 5e-6


@pytest.mark.parametrize('rows,c', [(50, 8), (300, 64), (77, 128), (40, 256), (9, 1024), (33, 96)])
def test_layernorm(ops, rows, c):
    from oracle.unet_ref import channel_layernorm
    x, gam = g((rows, c), 30, 1.5) + 0.3, g((c,), 31) * 0.2 + 1
    xr, gr = x.clone().requires_grad_(True), gam.clone().requires_grad_(True)
    yr = channel_layernorm(xr.t()[None], gr.reshape(1, c, 1))[0].t()
    go = g((rows, c), 32)
    yr.backward(go)
    xd, gd = dev(x, True), dev(gam.reshape(1, c, 1, 1), True)
    y = ops.layernorm_cl(xd, gd)
    assert rel_l2(y.detach(), yr.detach()) < TOL
    y.backward(dev(go))
    assert rel_l2(xd.grad, xr.grad) < 5e-6 and rel_l2(gd.grad.reshape(-1), gr.grad) < 5e-6


# ----------------------------------------------------------------------------------------------------- attention
def test_softmax_attention_thread_per_row_kernels(ops):
    """n_tok <= 32 normally takes the one-wave-per-item MFMA kernels; debug 5 forces the thread-per-row kernels (used for the
    mid-block spatial attention) on the same temporal cases."""
    lib = ops._lib_()
    lib.wdno_set_debug(5)
    try:
        test_softmax_attention(ops, 'temporal', 2, 24, 3, 5)
        test_softmax_attention(ops, 'temporal', 1, 7, 2, 2)
    finally:
        lib.wdno_set_debug(0)


@pytest.mark.parametrize('kind,b,f,h,w', [('temporal', 2, 24, 3, 5), ('temporal', 1, 7, 2, 2), ('spatial', 2, 3, 10, 10), ('spatial', 1, 1, 8, 8),
                                          ('temporal', 1, 48, 3, 4), ('temporal', 2, 33, 2, 3), ('temporal', 1, 56, 1, 2),
                                          ('spatial', 1, 2, 20, 20), ('spatial', 2, 1, 13, 11), ('spatial', 1, 1, 24, 24)])      # 400 / 143 / 576 tokens: attn_bwd_big_kernel
def test_softmax_attention(ops, kind, b, f, h, w):
    from oracle import unet_ref as U
    heads, dh = 4, 32
    qkv = g((b, f, h, w, 3 * heads * dh), 40)
    freqs = 1.0 / (10000 ** (torch.arange(0, 32, 2).double() / 32))
    qr = qkv.clone().requires_grad_(True)
    if kind == 'temporal':
        bias = g((heads, f, f), 41)
        br = bias.clone().requires_grad_(True)
        tok = qr.permute(0, 2, 3, 1, 4).reshape(b, h * w, f, -1)
        ntok = f
    else:
        bias, br = None, None
        tok = qr.reshape(b, f, h * w, -1)
        ntok = h * w
    q, k, v = tok.chunk(3, dim=-1)
    split = lambda z: z.reshape(*z.shape[:-1], heads, dh).transpose(-2, -3)
    q, k, v = split(q) * dh ** -0.5, split(k), split(v)
    if kind == 'temporal':
        q, k = U.rotary(q, freqs), U.rotary(k, freqs)
    sim = q @ k.transpose(-1, -2)
    if br is not None:
        sim = sim + br
    o = (sim.softmax(dim=-1) @ v).transpose(-2, -3).reshape(*tok.shape[:-1], heads * dh)
    outr = o.reshape(b, h, w, f, -1).permute(0, 3, 1, 2, 4) if kind == 'temporal' else o.reshape(b, f, h, w, -1)
    go = g(tuple(outr.shape), 42)
    outr.backward(go)

    qd = dev(qkv.reshape(-1, qkv.shape[-1]), True)
    if kind == 'temporal':
        bd = dev(bias, True)
        ang = (torch.arange(f).double()[:, None] * freqs[None]).repeat_interleave(2, dim=-1)
        rot = (dev(ang.cos()), dev(ang.sin()))
        out = ops.softmax_attention(qd, heads, b, h * w, f, f * h * w, 1, h * w, dh ** -0.5, bias=bd, rot=rot)
    else:
        out = ops.softmax_attention(qd, heads, b * f, 1, h * w, h * w, 0, 1, dh ** -0.5)
    assert rel_l2(out.detach().reshape(outr.shape), outr.detach()) < TOL
    out.backward(dev(go.reshape(-1, heads * dh)))
    assert rel_l2(qd.grad.reshape(qkv.shape), qr.grad) < 5e-6
    if kind == 'temporal':
        assert rel_l2(bd.grad, br.grad) < 5e-6


@pytest.mark.parametrize('units,n', [(3, 100), (2, 1600), (1, 37), (4, 16), (16, 4096), (2, 1000)])      # the last three pairs run with a chunked token range
def test_linear_attention(ops, units, n):
    heads, dh = 4, 32
    qkv = g((units, n, 3 * heads * dh), 50)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = (z.reshape(units, n, heads, dh).permute(0, 2, 3, 1) for z in qr.chunk(3, dim=-1))   # b h d n
    q = q.softmax(dim=-2) * dh ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum('bhdn,bhen->bhde', k, v)
    outr = torch.einsum('bhde,bhdn->bhen', ctx, q).permute(0, 3, 1, 2).reshape(units, n, heads * dh)
    go = g(tuple(outr.shape), 51)
    outr.backward(go)
    qd = dev(qkv.reshape(-1, qkv.shape[-1]), True)
    out = ops.linear_attention(qd, units, n, heads, dh ** -0.5)
    assert rel_l2(out.detach().reshape(outr.shape), outr.detach()) < TOL
    out.backward(dev(go.reshape(-1, heads * dh)))
    assert rel_l2(qd.grad.reshape(qkv.shape), qr.grad) < 5e-6


# ----------------------------------------------------------------------------------------------------- wavelets
G = load_npz('dwt_pywt.npz')


def _dw():
    from wdno_amd import wavelets as W
    return W


@pytest.mark.parametrize('tag', ['ref', 'small', 'db4', 'sym4', 'haar', 'b13'])
def test_dwt2_periodization_vs_pywt(ops, tag):
    W = _dw()
    wave = str(G[f'dwt2per_{tag}_wave'])
    x = torch.from_numpy(G[f'dwt2per_{tag}_x'])
    packed = W.dwt_packed(dev(x), wave, 'periodization', 2).cpu().double()
    assert (packed[:, :, 0] - torch.from_numpy(G[f'dwt2per_{tag}_yl'])).abs().max() < 3e-6
    assert (packed[:, :, 1:] - torch.from_numpy(G[f'dwt2per_{tag}_yh'])).abs().max() < 3e-6
    yl, yh = W.DWTForward(J=1, wave=wave, mode='periodization')(dev(x))
    assert torch.equal(yl.cpu().double(), packed[:, :, 0]) and torch.equal(yh[0].cpu().double(), packed[:, :, 1:])   # layout is bit-exact
    rec = W.DWTInverse(wave=wave, mode='periodization')((dev(torch.from_numpy(G[f'idwt2per_{tag}_yl'])), [dev(torch.from_numpy(G[f'idwt2per_{tag}_yh']))]))
    assert (rec.cpu().double() - torch.from_numpy(G[f'idwt2per_{tag}_x'])).abs().max() < 3e-6


@pytest.mark.parametrize('tag', ['per', 'perodd', 'zero', 'zeroodd', 'zdb4'])
def test_dwt1_vs_pywt(ops, tag):
    W = _dw()
    wave, mode = str(G[f'dwt1_{tag}_wave']), str(G[f'dwt1_{tag}_mode'])
    lo, hi = W.DWT1DForward(J=1, wave=wave, mode=mode)(dev(torch.from_numpy(G[f'dwt1_{tag}_x'])))
    assert (lo.cpu().double() - torch.from_numpy(G[f'dwt1_{tag}_lo'])).abs().max() < 3e-6
    assert (hi[0].cpu().double() - torch.from_numpy(G[f'dwt1_{tag}_hi'])).abs().max() < 3e-6
    rec = W.DWT1DInverse(wave=wave, mode=mode)((dev(torch.from_numpy(G[f'idwt1_{tag}_lo'])), [dev(torch.from_numpy(G[f'idwt1_{tag}_hi']))]))
    assert (rec.cpu().double() - torch.from_numpy(G[f'idwt1_{tag}_x'])).abs().max() < 3e-6


@pytest.mark.parametrize('tag', ['ref', 'odd'])
def test_dwt2_zero_vs_pywt(ops, tag):
    W = _dw()
    yl, yh = W.DWTForward(J=1, wave=str(G[f'dwt2zero_{tag}_wave']), mode='zero')(dev(torch.from_numpy(G[f'dwt2zero_{tag}_x'])))
    assert (yl.cpu().double() - torch.from_numpy(G[f'dwt2zero_{tag}_yl'])).abs().max() < 3e-6
    assert (yh[0].cpu().double() - torch.from_numpy(G[f'dwt2zero_{tag}_yh'])).abs().max() < 3e-6


@pytest.mark.parametrize('tag', ['mid', 'odd', 'db2'])
def test_dwt3_vs_pywt(ops, tag):
    W = _dw()
    wave = str(G[f'dwt3_{tag}_wave'])
    packed = W.wavedec3_packed(dev(torch.from_numpy(G[f'dwt3_{tag}_x'])), wave)
    assert (packed.cpu().double() - torch.from_numpy(G[f'dwt3_{tag}_coef'])).abs().max() < 3e-6
    c = torch.from_numpy(G[f'idwt3_{tag}_coef'])
    rec = W.waverec3([dev(c[:, 0]), {k: dev(c[:, i + 1]) for i, k in enumerate(W.BANDS3[1:])}], wave)
    assert (rec.cpu().double() - torch.from_numpy(G[f'idwt3_{tag}_x'])).abs().max() < 3e-6


@pytest.mark.parametrize('tag', ['ref3', 'even2', 'db4odd'])
def test_wavedec2_multilevel_vs_pywt(ops, tag):
    """J > 1 (burgers/wave_trans.py:94 builds DWTForward(J=dwt_max_level)); bands finest first, odd sizes cropped on the way back."""
    W = _dw()
    wave, J = str(G[f'wavedec2_{tag}_wave']), int(G[f'wavedec2_{tag}_J'])
    yl, yh = W.DWTForward(J=J, wave=wave, mode='periodization')(dev(torch.from_numpy(G[f'wavedec2_{tag}_x'])))
    assert len(yh) == J
    assert (yl.cpu().double() - torch.from_numpy(G[f'wavedec2_{tag}_yl'])).abs().max() < 5e-6
    for lvl in range(J):
        assert (yh[lvl].cpu().double() - torch.from_numpy(G[f'wavedec2_{tag}_yh{lvl}'])).abs().max() < 5e-6
    rec = W.DWTInverse(wave=wave, mode='periodization')((dev(torch.from_numpy(G[f'waverec2_{tag}_yl'])),
                                                         [dev(torch.from_numpy(G[f'waverec2_{tag}_yh{lvl}'])) for lvl in range(J)]))
    assert (rec.cpu().double() - torch.from_numpy(G[f'waverec2_{tag}_x'])).abs().max() < 1e-5


@pytest.mark.parametrize('tag', ['per3', 'zero2'])
def test_wavedec1_multilevel_vs_pywt(ops, tag):
    W = _dw()
    wave, mode, J = str(G[f'wavedec1_{tag}_wave']), str(G[f'wavedec1_{tag}_mode']), int(G[f'wavedec1_{tag}_J'])
    lo, his = W.DWT1DForward(J=J, wave=wave, mode=mode)(dev(torch.from_numpy(G[f'wavedec1_{tag}_x'])))
    assert (lo.cpu().double() - torch.from_numpy(G[f'wavedec1_{tag}_lo'])).abs().max() < 5e-6
    for lvl in range(J):
        assert (his[lvl].cpu().double() - torch.from_numpy(G[f'wavedec1_{tag}_hi{lvl}'])).abs().max() < 5e-6
    rec = W.DWT1DInverse(wave=wave, mode=mode)((lo, his))
    assert (rec.cpu().double() - torch.from_numpy(G[f'waverec1_{tag}_x'])).abs().max() < 1e-5


@pytest.mark.parametrize('tag', ['l2', 'l2odd'])
def test_wavedec3_multilevel_vs_pywt(ops, tag):
    W = _dw()
    wave, J = str(G[f'wavedec3_{tag}_wave']), int(G[f'wavedec3_{tag}_J'])
    c = W.wavedec3(dev(torch.from_numpy(G[f'wavedec3_{tag}_x'])), wave, mode='zero', level=J)
    assert len(c) == J + 1
    assert (c[0].cpu().double() - torch.from_numpy(G[f'wavedec3_{tag}_lll'])).abs().max() < 5e-6
    for lvl in range(J):
        d = torch.from_numpy(G[f'wavedec3_{tag}_d{lvl}'])
        assert list(c[1 + lvl].keys()) == list(W.BANDS3[1:])
        for i, k in enumerate(W.BANDS3[1:]):
            assert (c[1 + lvl][k].cpu().double() - d[:, i]).abs().max() < 5e-6
    rec = W.waverec3(c, wave)
    assert (rec.cpu().double() - torch.from_numpy(G[f'waverec3_{tag}_x'])).abs().max() < 1e-5


def test_multilevel_gradients_are_adjoints(ops):
    """<DWT_J x, c> == <x, DWT_J^T c> through autograd over two levels (guidance back-propagates through the IDWT)."""
    W = _dw()
    x = dev(torch.randn(2, 2, 32, 48)).requires_grad_(True)
    yl, yh = W.DWTForward(J=2, wave='bior2.4', mode='periodization')(x)
    cl, ch = torch.randn_like(yl), [torch.randn_like(h) for h in yh]
    lhs = (yl * cl).sum() + sum((h * c).sum() for h, c in zip(yh, ch))
    lhs.backward()
    cl.requires_grad_(True)
    rec = W.DWTInverse(wave='bior2.4', mode='periodization')((cl, ch))
    # bior2.4 is biorthogonal: analysis^T != synthesis, so check the adjoint identity numerically instead
    v = torch.randn_like(x)
    num = ((W.DWTForward(J=2, wave='bior2.4', mode='periodization')(v)[0] * cl).sum()
           + sum((h * c).sum() for h, c in zip(W.DWTForward(J=2, wave='bior2.4', mode='periodization')(v)[1], ch)))
    assert abs(float((x.grad * v).sum()) - float(num)) < 2e-3 * max(1.0, abs(float(num)))
    g = torch.randn_like(rec)
    (rec * g).sum().backward()
    assert cl.grad is not None and torch.isfinite(cl.grad).all()


@pytest.mark.parametrize('nd,mode,wave,shape', [(2, 'periodization', 'bior2.4', (3, 2, 81, 120)), (2, 'periodization', 'bior2.4', (2, 2, 160, 128)),
                                                (3, 'zero', 'bior1.3', (4, 32, 64, 64)), (3, 'zero', 'bior1.3', (2, 9, 11, 13)),
                                                (1, 'zero', 'bior1.3', (5, 33)), (1, 'periodization', 'db4', (4, 31)), (2, 'zero', 'bior1.3', (2, 17, 20))])
def test_dwt_vs_oracle_and_adjoints(ops, nd, mode, wave, shape):
    """fp32 HIP vs the fp32 numpy oracle, perfect reconstruction, and <A x, y> == <x, A^T y> for both autograd adjoints."""
    from oracle import dwt_ref as R
    W = _dw()
    x = g(shape, 60).float()
    xd = dev(x, True)
    packed = W.dwt_packed(xd, wave, mode, nd)
    xn = x.numpy()
    if nd == 1:
        lo, hi = R.dwt1d(xn[:, None], wave, mode)
        ref = np.stack([lo[:, 0], hi[:, 0]], axis=1)
    elif nd == 2:
        x4 = xn if xn.ndim == 4 else xn[:, None]
        yl, yh = R.dwt2(x4, wave, mode)
        ref = np.concatenate([yl[:, :, None], yh], axis=2)
        ref = ref if xn.ndim == 4 else ref[:, 0]
    else:
        lll, det = R.dwt3(xn, wave, mode)
        ref = R.smoke_coef_to_tensor(lll, det)
    assert packed.shape == ref.shape
    assert rel_l2(packed.detach(), torch.from_numpy(ref)) < 2e-6
    y = g(tuple(packed.shape), 61).float()
    (packed * dev(y)).sum().backward()
    lhs = (packed.detach().cpu().double() * y.double()).sum()
    rhs = (x.double() * xd.grad.cpu().double()).sum()
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), 1.0), 'analysis adjoint'
    cd = dev(y, True)
    rec = W.idwt_packed(cd, wave, mode, nd)
    z = g(tuple(rec.shape), 62).float()
    (rec * dev(z)).sum().backward()
    lhs = (rec.detach().cpu().double() * z.double()).sum()
    rhs = (y.double() * cd.grad.cpu().double()).sum()
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), 1.0), 'synthesis adjoint'
    back = W.idwt_packed(packed.detach(), wave, mode, nd).cpu()
    crop = tuple(slice(0, s) for s in shape[-nd:])
    assert (back[(Ellipsis, *crop)] - x).abs().max() < 1e-4, 'perfect reconstruction'


def test_upsample_coef(ops):
    import sys
    from wdno_amd import tree_path
    from oracle import dwt_ref as R
    a = g((2, 3, 5, 7), 70).float()
    out = ops.upsample_coef_raw(a.to(DEV), 6, 5, 1, 1, 7, 2, 1, 2).reshape(2, 3, 10, 14)
    assert torch.equal(out.cpu(), torch.from_numpy(R.upsample_coef_2d(a.numpy())))
    b = g((2, 4, 3, 5, 5), 71).float()
    t = ops.upsample_coef_raw(b.to(DEV), 2, 4, 75, 1, 1, 2, 1, 1).reshape(2, 8, 3, 5, 5)
    s = ops.upsample_coef_raw(b.to(DEV), 24, 5, 1, 1, 5, 2, 1, 2).reshape(2, 4, 3, 10, 10)
    assert torch.equal(t.cpu(), torch.from_numpy(R.upsample_coef_3d(b.numpy(), 'time')))
    assert torch.equal(s.cpu(), torch.from_numpy(R.upsample_coef_3d(b.numpy(), 'space')))


# ----------------------------------------------------------------------------------------------------- diffusion elementwise
def test_q_sample_cond_smoke_and_burgers(ops):
    from oracle import diffusion_ref as D
    from wdno_amd import diffusion_core as K
    buf = D.make_buffers('sigmoid', 1000)
    x0, nz = g((2, 6, 42, 8, 8), 80).float(), g((2, 6, 42, 8, 8), 81).float()
    t = torch.tensor([3, 900])
    for ctrl, pad in ((True, True), (False, True), (True, False)):
        xr = D.q_sample(buf, x0, t, nz)
        tr = nz.clone()
        kw = dict(is_condition_control=ctrl, is_condition_pad=pad)
        D.smoke_apply_conditions(xr, (4, 6, 5), init=x0[:, :, -2], control=x0[:, :, 24:40], **kw)
        D.smoke_apply_conditions(tr, (4, 6, 5), init=tr[:, :, -2], control=tr[:, :, 24:40], zero=True, **kw)
        desc = K.cond_desc(0, tuple(x0.shape), (4, 6, 5), pad, ctrl, 0, 0, False)
        xd, td = K.q_sample_cond(x0.to(DEV), nz.to(DEV), t.to(DEV), buf['sqrt_alphas_cumprod'].to(DEV), buf['sqrt_one_minus_alphas_cumprod'].to(DEV), desc)
        assert (xd.cpu() - xr).abs().max() < 1e-6 and torch.equal(td.cpu(), tr)
    bufb = D.make_buffers('cosine', 1000)
    x0, nz = g((2, 9, 16, 12), 82).float(), g((2, 9, 16, 12), 83).float()
    for flags in (dict(pad=True, u0=True, uT=False, f=True), dict(pad=True, u0=True, uT=True, f=True), dict(pad=False, u0=False, uT=True, f=False)):
        xr = D.q_sample(bufb, x0, t, nz)
        tr = nz.clone()
        D.burgers_apply_conditions(xr, (11, 10), flags, u0=x0[:, -1, :8], uT=x0[:, -1, 8:], f=x0[:, 4:8])
        z = torch.zeros_like
        D.burgers_apply_conditions(tr, (11, 10), flags, u0=z(x0[:, -1, :8]), uT=z(x0[:, -1, 8:]), f=z(x0[:, 4:8]))
        desc = K.cond_desc(1, tuple(x0.shape), (11, 10), flags['pad'], flags['u0'], flags['uT'], flags['f'], False, 8, 8)
        xd, td = K.q_sample_cond(x0.to(DEV), nz.to(DEV), t.to(DEV), bufb['sqrt_alphas_cumprod'].to(DEV), bufb['sqrt_one_minus_alphas_cumprod'].to(DEV), desc)
        assert (xd.cpu() - xr).abs().max() < 1e-6 and torch.equal(td.cpu(), tr)


def test_flat_adam_matches_torch(ops):
    from wdno_amd.trainer import FlatAdam
    torch.manual_seed(0)
    ps = [torch.randn(37, 5), torch.randn(101), torch.randn(3, 3, 3)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    mine = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
    opt_ref = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.99))
    opt = FlatAdam(mine, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0)
    for step in range(4):
        grads = [torch.randn_like(p) * (3.0 if step % 2 else 0.01) for p in ps]
        for r, m, gr in zip(ref, mine, grads):
            r.grad = gr.clone()
            m.grad.copy_(gr.to(DEV))
        gn_ref = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        opt_ref.step()
        gn = opt.step()
        assert abs(float(gn) - float(gn_ref)) < 1e-5 * max(1.0, float(gn_ref))
    for r, m in zip(ref, mine):
        assert rel_l2(m.detach(), r.detach()) < 1e-6


def test_relpos_bias_lookup_and_gradient(ops):
    """Relative-position bias table lookup (conv3d.py:106-112) and its gradient against torch's embedding (bit-exact forward; the
    backward sums at most 24 floats per table entry)."""
    torch.manual_seed(5)
    w = torch.randn(32, 4)
    idx = torch.randint(0, 32, (24, 24))
    wr = w.clone().requires_grad_(True)
    ref = wr[idx].permute(2, 0, 1)
    go = torch.randn(4, 24, 24)
    ref.backward(go)
    wd = w.to(DEV).requires_grad_(True)
    out = ops.relpos_bias(wd, idx.to(DEV))
    assert torch.equal(out.detach().cpu(), ref.detach())
    out.backward(go.to(DEV))
    assert rel_l2(wd.grad, wr.grad) < 1e-6


@pytest.mark.parametrize('shape,k,groups,use_ss', [((2, 64, 4, 16, 16), 64, 8, True), ((1, 128, 2, 24, 24), 128, 8, False), ((4, 128, 32, 32), 256, 1, True)])
def test_groupnorm_backward_delivers_planes(ops, shape, k, groups, use_ss):
    """conv -> GroupNorm(+scale/shift) -> SiLU with conv_cl(grad_planes=True): the norm's backward writes the fp16 planes of dx
    (scale from the analytic bound) and their column sums; every gradient must agree with the fp32-dx route to rounding."""
    nd = len(shape) - 2
    c = shape[1]
    x = g(shape, 301)
    w = g((k, c) + (3,) * nd, 302) * 0.05
    b = g((k,), 303) * 0.1
    gamma, beta = 1 + 0.2 * g((k,), 304), 0.1 * g((k,), 305)
    ss = 0.3 * g((shape[0], 2 * k), 306) if use_ss else None
    r = g((shape[0], k) + tuple(shape[2:]), 307)

    def run(flag):
        xs = dev(to_cl(x), grad=True)
        ws_, bs, gs, bes = dev(w, grad=True), dev(b, grad=True), dev(gamma, grad=True), dev(beta, grad=True)
        sss = dev(ss, grad=True) if use_ss else None
        calls = {'n': 0}
        orig = ops.split_f16_colsum

        def counting(*a, **kw):
            calls['n'] += 1
            return orig(*a, **kw)
        ops.split_f16_colsum = counting
        try:
            y = ops.conv_cl(xs, ws_, bs, padding=1, grad_planes=flag)
            z = ops.groupnorm_act(y, gs, bes, groups, sss, act=True)
            (z * dev(to_cl(r))).sum().backward()
        finally:
            ops.split_f16_colsum = orig
        out = [xs.grad, ws_.grad, bs.grad, gs.grad, bes.grad] + ([sss.grad] if use_ss else [])
        return [t.double().cpu() for t in out], calls['n']

    ref, n_ref = run(False)
    got, n_got = run(True)
    assert n_ref == 1 and n_got == 0, (n_ref, n_got)          # the planes route never splits dy
    for name, a, e in zip(['dx', 'dw', 'db', 'dgamma', 'dbeta', 'dss'], got, ref):
        assert rel_l2(a, e) < 2e-6, (name, rel_l2(a, e))


def test_norm_layers_write_planes_for_the_next_convolution(ops):
    """GroupNorm / LayerNorm with out_planes=True: the following convolution must produce the same result (to rounding) as on the
    fp32 output, in forward and in every gradient, without a split of its input."""
    x = g((2, 64, 4, 16, 16), 311)
    gamma, beta = 1 + 0.2 * g((64,), 312), 0.1 * g((64,), 313)
    ss = 0.3 * g((2, 128), 314)
    w = g((128, 64, 3, 3, 3), 315) * 0.05
    lg = 1 + 0.2 * g((1, 64, 1, 1, 1), 316)
    wq = g((384, 64), 317) * 0.1
    r1, r2 = g((2, 128, 4, 16, 16), 318), g((2, 384, 4, 16, 16), 319)

    def run(flag):
        calls = {'n': 0}
        orig = ops.split_f16

        def counting(*a, **kw):
            calls['n'] += 1
            return orig(*a, **kw)
        ops.split_f16 = counting
        try:
            xs = dev(to_cl(x), grad=True)
            gs, bes, sss, ws_, lgs, wqs = (dev(t, grad=True) for t in (gamma, beta, ss, w, lg, wq))
            assert ops.conv_reads_planes(2 * 4 * 16 * 16, ws_) and ops.conv_reads_planes(2 * 4 * 16 * 16, wqs)
            z = ops.groupnorm_act(xs, gs, bes, 8, sss, act=True, out_planes=flag)
            y1 = ops.conv_cl(z, ws_, None, padding=1)
            q = ops.layernorm_cl(xs, lgs, 1e-5, out_planes=flag)
            y2 = ops.conv_cl(q, wqs)
            ((y1 * dev(to_cl(r1))).sum() + (y2 * dev(to_cl(r2))).sum()).backward()
        finally:
            ops.split_f16 = orig
        outs = [y1.detach(), y2.detach(), xs.grad, gs.grad, bes.grad, sss.grad, ws_.grad, lgs.grad, wqs.grad]
        return [t.double().cpu() for t in outs], calls['n']

    ref, n_ref = run(False)
    got, n_got = run(True)
    assert n_got == n_ref - 2, (n_ref, n_got)           # the two forward splits are gone
    for name, a, e in zip(['y1', 'y2', 'dx', 'dgamma', 'dbeta', 'dss', 'dw', 'dlg', 'dwq'], got, ref):
        assert rel_l2(a, e) < 2e-6, (name, rel_l2(a, e))


def test_concat_as_planes_for_a_resnet_block(ops):
    """concat_cl(..., planes_only=True) in front of the two convolutions of a ResnetBlock input (3 x 3 x 3 and the 1 x 1 skip projection):
    same result and gradients (to rounding) as the fp32 concat, with one split pass less and no fp32 concat tensor."""
    a, b = g((2, 64, 4, 16, 16), 351), g((2, 64, 4, 16, 16), 352) * 3.0
    w3, w1 = g((64, 128, 3, 3, 3), 353) * 0.03, g((64, 128, 1, 1, 1), 354) * 0.1
    wa = g((64, 64, 1, 1, 1), 355) * 0.1
    r = g((2, 64, 4, 16, 16), 356)

    def run(flag):
        calls = {'n': 0}
        orig = ops.split_f16

        def counting(*a_, **kw):
            calls['n'] += 1
            return orig(*a_, **kw)
        ops.split_f16 = counting
        try:
            xa, xb, w3s, w1s, was = (dev(t, grad=True) for t in (to_cl(a), to_cl(b), w3, w1, wa))
            ya = ops.conv_cl(xa, was)                     # a producer that leaves an amax record on its output
            cat = ops.concat_cl(ya, ops.add(xb, xb), planes_only=flag)          # ops.add leaves a record too
            h, xs = ops.conv_cl_skip(cat, w3s, None, padding=1)
            y = ops.conv_cl(xs, w1s, None, residual=h)
            (y * dev(to_cl(r))).sum().backward()
        finally:
            ops.split_f16 = orig
        return [t.double().cpu() for t in (y.detach(), xa.grad, xb.grad, w3s.grad, w1s.grad, was.grad)], calls['n']
    ref, n_ref = run(False)
    got, n_got = run(True)
    assert n_got == n_ref - 1, (n_ref, n_got)
    for name, x_, e_ in zip(['y', 'da', 'db', 'dw3', 'dw1', 'dwa'], got, ref):
        assert rel_l2(x_, e_) < 2e-6, (name, rel_l2(x_, e_))


def test_attention_backward_delivers_planes(ops):
    """to_qkv -> temporal softmax attention -> to_out with conv_cl(grad_planes=True) on the qkv projection: the attention backward
    writes dqkv as fp16 planes (scale from the amax records of qkv and dout); gradients must agree with the fp32 route."""
    b, f, h, w, c, heads = 2, 8, 16, 16, 64, 4
    x = g((b, c, f, h, w), 321)
    wq = g((3 * heads * 32, c), 322) * 0.2
    wo = g((c, heads * 32), 323) * 0.1
    r = g((b, c, f, h, w), 324)

    def run(flag):
        calls = {'n': 0}
        orig = ops.split_f16

        def counting(*a, **kw):
            calls['n'] += 1
            return orig(*a, **kw)
        ops.split_f16 = counting
        try:
            xs, wqs, wos = dev(to_cl(x), grad=True), dev(wq, grad=True), dev(wo, grad=True)
            rows = ops.conv_cl(xs, wqs, grad_planes=flag)
            out = ops.softmax_attention(rows, heads, b, h * w, f, f * h * w, 1, h * w, 32 ** -0.5, out_planes=flag)
            y = ops.conv_cl(out, wos)
            (y * dev(to_cl(r))).sum().backward()
        finally:
            ops.split_f16 = orig
        return [t.double().cpu() for t in (xs.grad, wqs.grad, wos.grad)], calls['n']

    ref, n_ref = run(False)
    got, n_got = run(True)
    assert n_got == n_ref - 2, (n_ref, n_got)           # neither dqkv nor the attention output is split
    for name, a, e in zip(['dx', 'dwq', 'dwo'], got, ref):
        assert rel_l2(a, e) < 2e-6, (name, rel_l2(a, e))


def test_attention_two_tile_forward_writes_planes_when_sampling(ops):
    """33..64 tokens (the super-resolution model attends over 48 frames): the two-tile MFMA forward. Without gradients it may hand its
    output to the to_out projection as planes only; with gradients it must not (its backward reads the fp32 output)."""
    b, f, h, w, c, heads = 1, 48, 8, 8, 64, 4
    x = g((b, c, f, h, w), 341)
    wq = g((3 * heads * 32, c), 342) * 0.2
    wo = g((c, heads * 32), 343) * 0.1
    xs, wqs, wos = dev(to_cl(x)), dev(wq), dev(wo)

    def run(flag, grad):
        calls = {'n': 0}
        orig = ops.split_f16

        def counting(*a, **kw):
            calls['n'] += 1
            return orig(*a, **kw)
        ops.split_f16 = counting
        try:
            with torch.set_grad_enabled(grad):
                xin = xs.clone().requires_grad_(grad)
                rows = ops.conv_cl(xin, wqs)
                out = ops.softmax_attention(rows, heads, b, h * w, f, f * h * w, 1, h * w, 32 ** -0.5, out_planes=flag)
                y = ops.conv_cl(out, wos)
                if grad:
                    y.sum().backward()
        finally:
            ops.split_f16 = orig
        return y.detach().double().cpu(), calls['n']
    ref, n_ref = run(False, False)
    got, n_got = run(True, False)
    assert n_got == n_ref - 1 and rel_l2(got, ref) < 2e-6, (n_ref, n_got, rel_l2(got, ref))
    got_g, n_g = run(True, True)
    assert rel_l2(got_g, ref) < 2e-6


def test_linear_attention_backward_delivers_planes(ops):
    """to_qkv -> spatial linear attention -> to_out, dqkv as fp16 planes (scale bound from the records of qkv, dout and the measured dctx)."""
    b, f, h, w, c, heads = 2, 3, 16, 16, 64, 4
    x = g((b, c, f, h, w), 331)
    wq = g((3 * heads * 32, c), 332) * 0.3
    wo = g((c, heads * 32), 333) * 0.1
    r = g((b, c, f, h, w), 334)

    def run(flag):
        calls = {'n': 0}
        orig = ops.split_f16

        def counting(*a, **kw):
            calls['n'] += 1
            return orig(*a, **kw)
        ops.split_f16 = counting
        try:
            xs, wqs, wos = dev(to_cl(x), grad=True), dev(wq, grad=True), dev(wo, grad=True)
            qkv = ops.conv_cl(xs, wqs, grad_planes=flag)
            out = ops.linear_attention(qkv, b * f, h * w, heads, 32 ** -0.5, out_planes=flag)
            y = ops.conv_cl(out, wos)
            (y * dev(to_cl(r))).sum().backward()
        finally:
            ops.split_f16 = orig
        return [t.double().cpu() for t in (xs.grad, wqs.grad, wos.grad)], calls['n']

    ref, n_ref = run(False)
    got, n_got = run(True)
    assert n_got == n_ref - 2, (n_ref, n_got)
    for name, a, e in zip(['dx', 'dwq', 'dwo'], got, ref):
        assert rel_l2(a, e) < 2e-6, (name, rel_l2(a, e))
