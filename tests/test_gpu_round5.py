"""Round-5 GPU tests.

* The 7 x 7 x 7 stem with structural-zero hints (smoke/ddpm/diffusion_2d.py:1008-1033 writes the zero padding of the
  coefficient channels, video_diffusion_pytorch_conv3d.py:393 is the convolution): the launch that skips the all-zero reduction stages must
  return the bits of the launch that runs every stage; the hint rides on the tensor object and dies with any in-place change.
* The same kernel's two-level sums: the stem is where the split path had picked up its excess over torch's fp32.
GPU box only."""
import sys

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture()
def ops():
    from wdno_amd import ops as o
    o._lib_()
    yield o
    o.ZERO_BOX = True
    o.CONV_MATH = 'f16x3'
    o._lib_().wdno_set_debug(0)


@pytest.fixture(scope='module')
def trees():
    from wdno_amd import tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    from video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from ddpm.diffusion_2d import GaussianDiffusion as GD2
    return dict(Unet3D=Unet3D_with_Conv3D, GD2=GD2)


def _boxed_input(n, f, h, w, c, box, seed):
    """CL tensor [n, f, h, w, pad4(c)] whose channels < box[0] are zero wherever frame >= box[1] or row >= box[2] or column >= box[3]."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, f, h, w, c, generator=g)
    ch, f0, h0, w0 = box
    x[:, f0:, :, :, :ch] = 0
    x[:, :, h0:, :, :ch] = 0
    x[:, :, :, w0:, :ch] = 0
    cp = (c + 3) // 4 * 4
    return F.pad(x, (0, cp - c))


@pytest.mark.parametrize('math', ['f16x3', 'bf16'])
@pytest.mark.parametrize('shape,box', [((2, 24, 40, 40, 42), (40, 18, 34, 34)),        # the bench tensor and its padded_shape
                                       ((1, 24, 40, 40, 42), (40, 18, 34, 34)),
                                       ((2, 12, 40, 40, 42), (40, 5, 9, 40)),           # another box, no column padding (tiles of 256 pixels straddle the 1 600-pixel frames)
                                       ((1, 48, 80, 80, 82), (40, 34, 66, 66))])        # the space-SR tensor: six 16-channel blocks, two of them boxed
def test_stem_skipping_zero_stages_is_bit_identical(ops, math, shape, box):
    """conv_cl on a tensor that carries the hint (skips stages) == the same launch told to run every stage (library debug mode 57) == the
    launch without a hint (skips only the stages whose source lies outside the grid for a whole tile); and all of them are the exact
    convolution to fp32 accuracy (two-level sums: 5e-7 where the round-4 kernel measured 1.4e-6)."""
    n, f, h, w, c = shape
    ops.CONV_MATH = math
    x = _boxed_input(n, f, h, w, c, box, seed=sum(shape)).to(DEV)
    gw = torch.Generator().manual_seed(5)
    wt = (torch.randn(64, c, 7, 7, 7, generator=gw) / (c * 343) ** 0.5).to(DEV)
    b = torch.randn(64, generator=gw).to(DEV)
    lib = ops._lib_()

    def run(hint, debug):
        xi = x.clone()
        if hint:
            ops.set_zero_box(xi, box)
        lib.wdno_set_debug(debug)
        ops.PROFILE = {}
        try:
            y = ops.conv_cl(xi, wt, b, padding=3)
            torch.cuda.synchronize()
        finally:
            used, ops.PROFILE = set(ops.PROFILE), None
            lib.wdno_set_debug(0)
        if not (math == 'bf16' and ((c + 7) // 8 * 8) % 16):      # (one bf16 plane is padded to 8 channels, not to whole 16-channel blocks: 82 -> 88 runs on the chunked kernel)
            assert any('h3t' in k for k in used), used
        return y
    y_all, y_oob, y_box = run(False, 57), run(False, 0), run(True, 0)
    assert torch.equal(y_all, y_oob) and torch.equal(y_all, y_box)
    if math == 'f16x3':
        ref = F.conv3d(x[..., :c].permute(0, 4, 1, 2, 3).double().cpu(), wt.double().cpu(), b.double().cpu(), padding=3).permute(0, 2, 3, 4, 1)
        assert rel_l2(y_box.cpu()[..., :64], ref) < 5e-7


def test_zero_box_hint_dies_with_an_in_place_change(ops):
    x = torch.zeros(1, 2, 4, 4, 8, device=DEV)
    ops.set_zero_box(x, (8, 1, 2, 2))
    assert ops.zero_box_of(x) == (8, 1, 2, 2)
    y = torch.empty_like(x)
    ops.carry_zero_box(y, x)
    assert ops.zero_box_of(y) == (8, 1, 2, 2)
    x.add_(1.0)
    assert ops.zero_box_of(x) is None and ops.zero_box_of(y) == (8, 1, 2, 2)
    ops.ZERO_BOX = False
    assert ops.zero_box_of(y) is None


def test_training_step_and_sampling_do_not_change_with_the_hint(ops, trees):
    """Full-width smoke model at [2, 24, 42, 40, 40]: loss, every gradient and two DDPM sampling steps with the hints reaching the stem
    (p_losses -> q_sample_cond; p_sample -> apply_cond) are the bits of the run with ops.ZERO_BOX = False."""
    from wdno_amd import diffusion_core as K

    def run(flag):
        ops.ZERO_BOX = flag
        torch.manual_seed(0)
        net = trees['Unet3D'](dim=64, dim_mults=(1, 2, 4), channels=42)
        dif = trees['GD2'](net, torch.ones(1, 1, 42, 1, 1), True, True, True, False, 'bior1.3', 'zero', (18, 34, 34), (32, 64, 64),
                           image_size=40, frames=24).to(DEV)
        g = torch.Generator().manual_seed(3)
        x0 = (torch.randn(2, 24, 42, 40, 40, generator=g) * 0.5).to(DEV)
        noise = torch.randn(2, 24, 42, 40, 40, generator=g).to(DEV)
        t = torch.tensor([37, 911], device=DEV)
        loss = dif.p_losses(x0, t, noise=noise)
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in dif.named_parameters() if p.grad is not None}
        shape = (2, 24, 42, 40, 40)
        init = torch.randn(2, 24, 40, 40, generator=g).to(DEV)
        control = torch.randn(2, 24, 16, 40, 40, generator=g).to(DEV)
        desc = dif._desc(shape, dif.padded_shape)
        src = dif._condition_source(shape, torch.device(DEV), init, control, None)
        seq = iter([torch.randn(shape, generator=g).to(DEV) for _ in range(2)])
        dif.sample_noise = lambda shp, device: next(seq)
        with torch.no_grad():
            x = K.apply_cond(x0.clone(), src, desc)
            assert (ops.zero_box_of(x) == (40, 18, 34, 34)) == flag
            for tt in (500, 499):
                x, _ = dif.p_sample(shape, x, tt)
                x = K.apply_cond(x, src, desc)
        torch.cuda.synchronize()
        return loss.detach().clone(), grads, x.clone()
    l0, g0, x0_ = run(False)
    l1, g1, x1_ = run(True)
    assert torch.equal(l0, l1) and torch.equal(x0_, x1_)
    assert g0.keys() == g1.keys() and all(torch.equal(g0[k], g1[k]) for k in g0)


def test_sampling_step_kernel_selection_full_size():
    """One p_sample step of the smoke base model at the bench size [8,24,42,40,40] under no_grad: which timed launches it is made of. Every
    temporal attention block (four at 64 channels, four at 128 / 256: csrc/attn_fused.hip, attn_fused_wide.hip) and every linear attention
    block (three + three: linattn_fused.hip, linattn_fused_wide.hip) is ONE profiled launch group; the only layer-by-layer attention left is
    the mid spatial block -- a silent fall-back to LayerNorm -> to_qkv -> attention -> to_out chains would show up here, not only as a timing."""
    import os
    from tests.helpers import GOLDEN
    sys.path.insert(0, os.path.dirname(GOLDEN.rstrip('/')).rsplit('/tests', 1)[0])
    import bench
    from wdno_amd import ops
    dev = torch.device('cuda', 0)
    dif = bench.build_model(dev, 8)
    shape = (8, 24, 42, 40, 40)
    x = torch.randn(shape, device=dev)
    with torch.no_grad():
        x, _ = dif.p_sample(shape, x, 500)
        ops.PROFILE = {}
        try:
            dif.p_sample(shape, x, 499)
            torch.cuda.synchronize()
            got = {k: len(v) for k, v in ops.PROFILE.items()}
        finally:
            ops.PROFILE = None
    assert got.get('tattn_fused_fwd_kernel') == 8 and got.get('lattn_fused_fwd_kernels') == 6, got
    assert not any(k.startswith('tattn_fused_bwd') or k.startswith('lattn_fused_bwd') or 'wgrad' in k for k in got), got
    # 1 x 1 projections that remain: the mid spatial block's to_qkv / to_out, the ResnetBlocks' res_conv, the output head
    assert got.get('conv_fwd_h3d_kernel<192,128>', 0) + got.get('conv_fwd_h3d_kernel<128,128>', 0) <= 6, got
    ops.drop_weight_caches()
    torch.cuda.empty_cache()
