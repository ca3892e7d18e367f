"""csrc/linattn_fused.hip: Residual(PreNorm(SpatialLinearAttention)) of the smoke U-Net's 64-channel levels (conv3d.py:165-174, 232-258) as two
passes over the tokens + a merge, without the [pixels x 384] projections -- the form sampling runs. Checked against the oracle's restatement
of the block in fp64 (the arbiter), with the layer-by-layer HIP path and the fp32 oracle as yardsticks, and inside the whole U-Net forward."""
import sys

import pytest
import torch

from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def mods():
    from wdno_amd import ops, tree_path
    for t in ('third_party', 'smoke', 'burgers'):
        p = tree_path(t)
        if p not in sys.path:
            sys.path.insert(0, p)
    from video_diffusion_pytorch import video_diffusion_pytorch_conv3d as V
    return ops, V


def _block(V, seed):
    torch.manual_seed(seed)
    blk = V.Residual(V.PreNorm(64, V.SpatialLinearAttention(64, heads=4)))
    with torch.no_grad():
        blk.fn.norm.gamma.add_(0.3 * torch.randn_like(blk.fn.norm.gamma))
        blk.fn.fn.to_qkv.weight.mul_(3.0)            # wider logits: the softmax over 1 600 tokens is not flat
    return blk


def _oracle(blk, x_cl, dt):
    from oracle import unet_ref as U
    att = blk.fn.fn
    x = x_cl.to(dt).permute(0, 4, 1, 2, 3)                                   # b c f h w
    y = U.channel_layernorm(x, blk.fn.norm.gamma.detach().to(dt))
    b, c, f, hh, ww = y.shape
    y = y.permute(0, 2, 1, 3, 4).reshape(b * f, c, hh, ww)
    y = U.linear_attention_2d(y, att.to_qkv.weight.detach().to(dt), att.to_out.weight.detach().to(dt), att.to_out.bias.detach().to(dt), 4, 32)
    return (y.reshape(b, f, c, hh, ww).permute(0, 2, 1, 3, 4) + x).permute(0, 2, 3, 4, 1)


@pytest.mark.parametrize('b,f,h,w', [(1, 2, 8, 8), (2, 3, 7, 11), (1, 5, 20, 20), (2, 24, 40, 40)])
def test_fused_linear_attention_block_vs_oracle_and_layers(mods, b, f, h, w):
    ops, V = mods
    blk = _block(V, 5)
    x = torch.randn(b, f, h, w, 64) * 1.5 + 0.2
    exact = _oracle(blk, x, torch.float64)
    ref32 = _oracle(blk, x, torch.float32)
    blk = blk.to(DEV)
    xd = x.to(DEV)
    with torch.no_grad():
        assert ops.lattn_fused_takes(xd, 4, (blk.fn.norm.gamma,))
        ops.PROFILE = {}
        y = blk(xd)
        used = set(ops.PROFILE)
        ops.PROFILE = None
        assert 'lattn_fused_fwd_kernels' in used and not any('conv' in k or 'linattn' in k for k in used), used
        assert ops._known_amax(y) is not None and abs(ops._known_amax(y).max().item() - y.abs().max().item()) == 0.0
        ops.FUSED_LATTN = False
        try:
            y_layers = blk(xd)
        finally:
            ops.FUSED_LATTN = True
    e_f, e_l, e_r = rel_l2(y, exact), rel_l2(y_layers, exact), rel_l2(ref32, exact)
    # the attention branch alone (y - x), where the residual does not mask the error
    b_f = rel_l2(y.cpu().double() - x.double(), exact - x.double())
    b_l = rel_l2(y_layers.cpu().double() - x.double(), exact - x.double())
    print(f'fused linear attention [{b},{f},{h},{w},64]: fused vs exact {e_f:.2e} (branch {b_f:.2e}), layer by layer {e_l:.2e} (branch {b_l:.2e}), fp32 oracle {e_r:.2e}')
    assert e_f < 1e-6 and e_f <= 1.5 * max(e_l, e_r) + 1e-7
    assert b_f < 5e-6 and b_f <= 1.5 * b_l + 5e-7


def _block_c(V, seed, c):
    torch.manual_seed(seed)
    blk = V.Residual(V.PreNorm(c, V.SpatialLinearAttention(c, heads=4)))
    with torch.no_grad():
        blk.fn.norm.gamma.add_(0.3 * torch.randn_like(blk.fn.norm.gamma))
        blk.fn.fn.to_qkv.weight.mul_(3.0)
    return blk


@pytest.mark.parametrize('c,b,f,h,w', [(128, 1, 2, 8, 8), (128, 2, 3, 7, 11), (128, 2, 24, 20, 20), (256, 1, 3, 6, 6), (256, 2, 5, 9, 5), (256, 2, 24, 10, 10)])
def test_wide_linear_attention_block_forward(mods, c, b, f, h, w):
    """csrc/linattn_fused_wide.hip: the block of the 128- / 256-channel levels (weights streamed in fragment order), forward only: against
    the fp64 oracle and the layer-by-layer HIP path; bit-reproducible; not taken when a gradient is needed."""
    ops, V = mods
    blk = _block_c(V, 13, c)
    x = torch.randn(b, f, h, w, c) * 1.5 + 0.2
    exact = _oracle(blk, x, torch.float64)
    ref32 = _oracle(blk, x, torch.float32)
    blk = blk.to(DEV)
    xd = x.to(DEV)
    with torch.no_grad():
        assert ops.lattn_fused_takes(xd, 4, (blk.fn.norm.gamma,))
        ops.PROFILE = {}
        y = blk(xd)
        used = set(ops.PROFILE)
        ops.PROFILE = None
        assert 'lattn_fused_fwd_kernels' in used and not any('conv' in k or 'linattn' in k for k in used), used
        assert ops._known_amax(y) is not None and abs(ops._known_amax(y).max().item() - y.abs().max().item()) == 0.0
        assert torch.equal(y, blk(xd))
        ops.FUSED_LATTN_WIDE = False
        try:
            assert not ops.lattn_fused_takes(xd, 4, (blk.fn.norm.gamma,))
            y_layers = blk(xd)
        finally:
            ops.FUSED_LATTN_WIDE = True
    e_f, e_l, e_r = rel_l2(y, exact), rel_l2(y_layers, exact), rel_l2(ref32, exact)
    b_f = rel_l2(y.cpu().double() - x.double(), exact - x.double())
    b_l = rel_l2(y_layers.cpu().double() - x.double(), exact - x.double())
    print(f'wide linear attention [{b},{f},{h},{w},{c}]: fused vs exact {e_f:.2e} (branch {b_f:.2e}), layer by layer {e_l:.2e} (branch {b_l:.2e}), fp32 oracle {e_r:.2e}')
    assert e_f < 1e-6 and e_f <= 1.5 * max(e_l, e_r) + 1e-7
    assert b_f < 5e-6 and b_f <= 1.5 * b_l + 5e-7
    xg = xd.clone().requires_grad_(True)
    assert not ops.lattn_fused_takes(xg, 4, (blk.fn.norm.gamma,))          # a gradient: layer by layer


def test_unet_forward_uses_the_fused_linear_attention_when_sampling(mods):
    """The whole denoiser under no_grad: the three 64-channel SpatialLinearAttention blocks (downs[0], ups[1], ups[2]) and the 128-channel one
    of the second level (csrc/linattn_fused_wide.hip) run fused; result vs the oracle and vs the layer-by-layer path; with gradients the
    64-channel blocks stay fused, the wide one runs layer by layer."""
    ops, V = mods
    from oracle import unet_ref as U
    torch.manual_seed(4)
    net = V.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x, t = torch.randn(1, 24, 42, 16, 16) * 0.7, torch.tensor([611])
    with torch.no_grad():
        ref = U.unet3d_forward(sd, x, t, dim=64, dim_mults=(1, 2, 4), groups=8)
        ref64 = U.unet3d_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x.double(), t, dim=64, dim_mults=(1, 2, 4), groups=8)
    net = net.to(DEV)
    try:
        with torch.no_grad():
            ops.PROFILE = {}
            out = net(x.to(DEV), t.to(DEV))
            n_fused = len(ops.PROFILE.get('lattn_fused_fwd_kernels', []))
            ops.PROFILE = None
            ops.FUSED_TATTN_WIDE = ops.FUSED_LATTN_WIDE = False      # the blocks a pass with gradients can fuse
            ops.PROFILE = {}
            out_narrow = net(x.to(DEV), t.to(DEV))
            n_narrow = len(ops.PROFILE.get('lattn_fused_fwd_kernels', []))
            ops.PROFILE = None
            ops.FUSED_LATTN = False
            out_layers = net(x.to(DEV), t.to(DEV))
    finally:
        ops.FUSED_LATTN = ops.FUSED_TATTN_WIDE = ops.FUSED_LATTN_WIDE = True
    assert n_fused == 4, n_fused            # downs[0], ups[1], ups[2] (64 channels) + downs[1] (128 channels on 8 x 8 = 64 tokens: csrc/linattn_fused_wide.hip)
    assert n_narrow == 3, n_narrow
    e_f, e_n, e_l, e_r = rel_l2(out, ref64), rel_l2(out_narrow, ref64), rel_l2(out_layers, ref64), rel_l2(ref, ref64)
    print(f'U-Net forward: fused linear attention vs exact {e_f:.2e} (64-channel blocks only {e_n:.2e}), layers vs exact {e_l:.2e}, fp32 oracle vs exact {e_r:.2e}')
    assert rel_l2(out, ref) < 1e-5 and e_f <= 1.5 * max(e_l, e_r) + 1e-7 and e_n <= 1.5 * max(e_l, e_r) + 1e-7
    ops.PROFILE = {}
    out_g = net(x.to(DEV), t.to(DEV))       # parameters require gradients here: the 64-channel blocks still fused (the backward is fused too)
    assert len(ops.PROFILE.get('lattn_fused_fwd_kernels', [])) == 3
    ops.PROFILE = None
    assert torch.equal(out_g.detach(), out_narrow)
    ops.FUSED_LATTN_BWD = False
    try:
        ops.PROFILE = {}
        net(x.to(DEV), t.to(DEV))           # test knob: a step with gradients runs the block layer by layer
        assert 'lattn_fused_fwd_kernels' not in ops.PROFILE
        ops.PROFILE = None
    finally:
        ops.FUSED_LATTN_BWD = True


def _grads_of(blk, x, gy, dev):
    xr = x.detach().clone().to(dev).requires_grad_(True)
    for p_ in blk.parameters():
        p_.grad = None
    y = blk(xr)
    y.backward(gy.to(dev))
    att = blk.fn.fn
    g = {'gamma': blk.fn.norm.gamma.grad, 'to_qkv': att.to_qkv.weight.grad, 'to_out': att.to_out.weight.grad, 'b_out': att.to_out.bias.grad}
    return y.detach(), xr.grad.clone(), {k: v.clone() for k, v in g.items()}


@pytest.mark.parametrize('b,f,h,w', [(1, 2, 8, 8), (2, 3, 7, 11), (1, 5, 20, 20), (2, 24, 40, 40)])
def test_fused_linear_attention_backward_vs_oracle_and_layers(mods, b, f, h, w):
    """dx, dgamma, dW_qkv, dW_out, db_out of the fused block against the fp64 oracle's autograd (exact), the layer-by-layer HIP path and the
    fp32 oracle as yardsticks; two evaluations give the same bits."""
    ops, V = mods
    blk = _block(V, 9)
    torch.manual_seed(13)
    x = torch.randn(b, f, h, w, 64) * 1.5 + 0.2
    gy = torch.randn(b, f, h, w, 64) * (0.5 + torch.rand(b, f, 1, 1, 1) * 4.0)

    def oracle(dt):
        from oracle import unet_ref as U
        att = blk.fn.fn
        prm = {'gamma': blk.fn.norm.gamma, 'to_qkv': att.to_qkv.weight, 'to_out': att.to_out.weight, 'b_out': att.to_out.bias}
        leaf = {k: v.detach().to(dt).requires_grad_(True) for k, v in prm.items()}
        xe = x.detach().clone().to(dt).requires_grad_(True)
        xc = xe.permute(0, 4, 1, 2, 3)
        y = U.channel_layernorm(xc, leaf['gamma'])
        bb, c, ff, hh_, ww_ = y.shape
        y = y.permute(0, 2, 1, 3, 4).reshape(bb * ff, c, hh_, ww_)
        y = U.linear_attention_2d(y, leaf['to_qkv'], leaf['to_out'], leaf['b_out'], 4, 32)
        y = (y.reshape(bb, ff, c, hh_, ww_).permute(0, 2, 1, 3, 4) + xc).permute(0, 2, 3, 4, 1)
        y.backward(gy.to(dt))
        return xe.grad, {k: v.grad for k, v in leaf.items()}

    dx_e, g_e = oracle(torch.float64)
    dx_r, g_r = oracle(torch.float32)
    blk = blk.to(DEV)
    ops.PROFILE = {}
    y_f, dx_f, g_f = _grads_of(blk, x, gy, DEV)
    used = set(ops.PROFILE)
    ops.PROFILE = None
    assert 'lattn_fused_bwd_kernels' in used and not any('conv' in k or 'linattn' in k or 'layernorm' in k for k in used), used
    _, dx_f2, g_f2 = _grads_of(blk, x, gy, DEV)
    assert torch.equal(dx_f, dx_f2) and all(torch.equal(g_f[k], g_f2[k]) for k in g_f)          # no atomics: bit-reproducible
    ops.FUSED_LATTN_BWD = False
    try:
        y_l, dx_l, g_l = _grads_of(blk, x, gy, DEV)
    finally:
        ops.FUSED_LATTN_BWD = True
    print(f'fused linear attention backward [{b},{f},{h},{w},64]:')
    worst = 0.0
    rows = [('dx-dy', dx_f.cpu().double() - gy.double(), dx_l.cpu().double() - gy.double(), dx_r.double() - gy.double(), dx_e - gy.double())]
    rows += [(k, g_f[k], g_l[k], g_r[k], g_e[k]) for k in g_f]
    for name, f_, l_, r_, e_ in rows:
        e_f, e_l, e_r = rel_l2(f_, e_), rel_l2(l_, e_), rel_l2(r_, e_)
        print(f'  {name:8s} fused vs exact {e_f:.2e}   layer by layer vs exact {e_l:.2e}   fp32 oracle vs exact {e_r:.2e}')
        worst = max(worst, e_f / (1.5 * max(e_l, e_r) + 3e-7), e_f / 1e-5)
    assert worst <= 1.0, worst
