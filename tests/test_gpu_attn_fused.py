"""csrc/attn_fused.hip + csrc/attn_fused_bwd.hip: Residual(PreNorm(temporal attention)) of the smoke U-Net's first level (conv3d.py:165-174,
277-353) as one launch forward and one launch backward. Checked against the oracle's restatement of the block in fp64 (the arbiter; its
autograd gives the exact gradients), against the layer-by-layer HIP path, and inside the whole U-Net (forward when sampling, all
gradients of a training step)."""
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


def _block(V, seed, dim=64):
    torch.manual_seed(seed)
    rot = V.RotaryEmbedding(32)
    att = V.EinopsToAndFrom('b c f h w', 'b (h w) f c', V.Attention(dim, heads=4, dim_head=32, rotary_emb=rot))
    blk = V.Residual(V.PreNorm(dim, att))
    with torch.no_grad():
        blk.fn.norm.gamma.add_(0.3 * torch.randn_like(blk.fn.norm.gamma))
        att.fn.to_qkv.weight.mul_(2.0)
    rpb = V.RelativePositionBias(heads=4, max_distance=32)
    return blk, att.fn, rpb


def _oracle(blk, att, rpb, x_cl, dt):
    """x_cl [B, F, H, W, C] -> the block by oracle/unet_ref.py's functions in dtype dt."""
    from oracle import unet_ref as U
    x = x_cl.to(dt).permute(0, 4, 1, 2, 3)                                   # b c f h w
    bias = U.time_rel_pos_bias(rpb.relative_attention_bias.weight.detach().to(dt), x.shape[2])
    y = U.channel_layernorm(x, blk.fn.norm.gamma.detach().to(dt))
    b, c, f, hh, ww = y.shape
    y = y.permute(0, 3, 4, 2, 1).reshape(b, hh * ww, f, c)
    y = U.token_attention(y, att.to_qkv.weight.detach().to(dt), att.to_out.weight.detach().to(dt), 4, 32,
                          freqs=att.rotary_emb.freqs.detach().to(dt), pos_bias=bias)
    return (y.reshape(b, hh, ww, f, c).permute(0, 4, 3, 1, 2) + x).permute(0, 2, 3, 4, 1)


@pytest.mark.parametrize('b,h,w', [(1, 8, 8), (3, 7, 11), (2, 40, 40)])
def test_fused_block_vs_oracle_and_layers(mods, b, h, w):
    ops, V = mods
    blk, att, rpb = _block(V, 5)
    x = torch.randn(b, 24, h, w, 64) * 1.5 + 0.2
    exact = _oracle(blk, att, rpb, x, torch.float64)
    ref32 = _oracle(blk, att, rpb, x, torch.float32)
    blk, rpb = blk.to(DEV), rpb.to(DEV)
    xd = x.to(DEV)
    with torch.no_grad():
        bias = rpb(24, device=DEV)
        assert ops.tattn_fused_takes(xd, 4, (blk.fn.norm.gamma,))
        ops.PROFILE = {}
        y = blk(xd, pos_bias=bias)
        used = set(ops.PROFILE)
        ops.PROFILE = None
        assert 'tattn_fused_fwd_kernel' in used and not any('conv' in k for k in used), used      # one launch, no projection kernels
        assert ops._known_amax(y) is not None and abs(ops._known_amax(y).max().item() - y.abs().max().item()) == 0.0
        ops.FUSED_TATTN = False
        try:
            y_layers = blk(xd, pos_bias=bias)
        finally:
            ops.FUSED_TATTN = True
    e_f, e_l, e_r = rel_l2(y, exact), rel_l2(y_layers, exact), rel_l2(ref32, exact)
    print(f'fused block [{b},24,{h},{w},64]: fused vs exact {e_f:.2e}, layer by layer vs exact {e_l:.2e}, fp32 oracle vs exact {e_r:.2e}, fused vs layers {rel_l2(y, y_layers):.2e}')
    assert e_f < 1e-6 and e_f <= 1.5 * max(e_l, e_r) + 1e-7
    # the attention branch alone (y - x), where the residual does not mask the error
    assert rel_l2(y.cpu().double() - x.double(), exact - x.double()) < 5e-6


@pytest.mark.parametrize('c,b,h,w', [(128, 1, 8, 8), (128, 3, 7, 11), (128, 8, 20, 20), (256, 1, 8, 8), (256, 2, 9, 5), (256, 8, 10, 10)])
def test_wide_block_forward(mods, c, b, h, w):
    """csrc/attn_fused_wide.hip: the block of the 128- / 256-channel levels (weights streamed from L2), forward only: against the fp64 oracle
    and the layer-by-layer HIP path; with and without rotary / bias; bit-reproducible; not taken when a gradient is needed."""
    ops, V = mods
    blk, att, rpb = _block(V, 11, dim=c)
    x = torch.randn(b, 24, h, w, c) * 1.5 + 0.2
    exact = _oracle(blk, att, rpb, x, torch.float64)
    ref32 = _oracle(blk, att, rpb, x, torch.float32)
    blk, rpb = blk.to(DEV), rpb.to(DEV)
    xd = x.to(DEV)
    with torch.no_grad():
        bias = rpb(24, device=DEV)
        assert ops.tattn_fused_takes(xd, 4, (blk.fn.norm.gamma,))
        ops.PROFILE = {}
        y = blk(xd, pos_bias=bias)
        used = set(ops.PROFILE)
        ops.PROFILE = None
        assert 'tattn_fused_fwd_kernel' in used and not any('conv' in k for k in used), used
        assert ops._known_amax(y) is not None and abs(ops._known_amax(y).max().item() - y.abs().max().item()) == 0.0
        assert torch.equal(y, blk(xd, pos_bias=bias))
        ops.FUSED_TATTN_WIDE = False
        try:
            assert not ops.tattn_fused_takes(xd, 4, (blk.fn.norm.gamma,))
            y_layers = blk(xd, pos_bias=bias)
        finally:
            ops.FUSED_TATTN_WIDE = True
        y_plain = ops.temporal_attention_fused(xd, blk.fn.norm.gamma, blk.fn.norm.eps, att.to_qkv.weight, att.to_out.weight, None, None, 4, att.scale)
    e_f, e_l, e_r = rel_l2(y, exact), rel_l2(y_layers, exact), rel_l2(ref32, exact)
    print(f'wide block [{b},24,{h},{w},{c}]: fused vs exact {e_f:.2e}, layer by layer vs exact {e_l:.2e}, fp32 oracle vs exact {e_r:.2e}, fused vs layers {rel_l2(y, y_layers):.2e}')
    assert e_f < 1e-6 and e_f <= 1.5 * max(e_l, e_r) + 1e-7
    assert rel_l2(y.cpu().double() - x.double(), exact - x.double()) < 5e-6
    from oracle import unet_ref as U
    xe = x.double().permute(0, 4, 1, 2, 3)
    yn = U.channel_layernorm(xe, blk.fn.norm.gamma.detach().double().cpu())
    bb, cc, f, hh, ww = yn.shape
    yn = yn.permute(0, 3, 4, 2, 1).reshape(bb, hh * ww, f, cc)
    yn = U.token_attention(yn, att.to_qkv.weight.detach().double().cpu(), att.to_out.weight.detach().double().cpu(), 4, 32)
    plain = (yn.reshape(bb, hh, ww, f, cc).permute(0, 4, 3, 1, 2) + xe).permute(0, 2, 3, 4, 1)
    assert rel_l2(y_plain, plain) < 1e-6
    xg = xd.clone().requires_grad_(True)
    assert not ops.tattn_fused_takes(xg, 4, (blk.fn.norm.gamma,))          # a gradient: layer by layer


@pytest.mark.parametrize('b,h,w', [(1, 8, 8), (2, 7, 11), (2, 40, 20)])
def test_fused_block_48_frames_forward(mods, b, h, w):
    """csrc/attn_fused48.hip: the block on the 48 wavelet frames of the super-resolution model (two token tiles per head, eight waves per
    sequence), forward only: against the fp64 oracle and the layer-by-layer HIP path; with gradients enabled the block is not fused."""
    ops, V = mods
    blk, att, rpb = _block(V, 7)
    x = torch.randn(b, 48, h, w, 64) * 1.5 + 0.2
    exact = _oracle(blk, att, rpb, x, torch.float64)
    ref32 = _oracle(blk, att, rpb, x, torch.float32)
    blk, rpb = blk.to(DEV), rpb.to(DEV)
    xd = x.to(DEV)
    with torch.no_grad():
        bias = rpb(48, device=DEV)
        assert ops.tattn_fused_takes(xd, 4, (blk.fn.norm.gamma,))
        ops.PROFILE = {}
        y = blk(xd, pos_bias=bias)
        used = set(ops.PROFILE)
        ops.PROFILE = None
        assert 'tattn_fused_fwd_kernel' in used and not any('conv' in k for k in used), used
        assert ops._known_amax(y) is not None and abs(ops._known_amax(y).max().item() - y.abs().max().item()) == 0.0
        y2 = blk(xd, pos_bias=bias)
        assert torch.equal(y, y2)
        ops.FUSED_TATTN = False
        try:
            y_layers = blk(xd, pos_bias=bias)
        finally:
            ops.FUSED_TATTN = True
        # no rotary, no bias
        y_plain = ops.temporal_attention_fused(xd, blk.fn.norm.gamma, blk.fn.norm.eps, att.to_qkv.weight, att.to_out.weight, None, None, 4, att.scale)
    e_f, e_l, e_r = rel_l2(y, exact), rel_l2(y_layers, exact), rel_l2(ref32, exact)
    print(f'fused block [{b},48,{h},{w},64]: fused vs exact {e_f:.2e}, layer by layer vs exact {e_l:.2e}, fp32 oracle vs exact {e_r:.2e}')
    assert e_f < 1e-6 and e_f <= 1.5 * max(e_l, e_r) + 1e-7
    assert rel_l2(y.cpu().double() - x.double(), exact - x.double()) < 5e-6
    from oracle import unet_ref as U
    xe = x.double().permute(0, 4, 1, 2, 3)
    yn = U.channel_layernorm(xe, blk.fn.norm.gamma.detach().double().cpu())
    bb, c, f, hh, ww = yn.shape
    yn = yn.permute(0, 3, 4, 2, 1).reshape(bb, hh * ww, f, c)
    yn = U.token_attention(yn, att.to_qkv.weight.detach().double().cpu(), att.to_out.weight.detach().double().cpu(), 4, 32)
    plain = (yn.reshape(bb, hh, ww, f, c).permute(0, 4, 3, 1, 2) + xe).permute(0, 2, 3, 4, 1)
    assert rel_l2(y_plain, plain) < 1e-6
    xg = xd.clone().requires_grad_(True)
    assert not ops.tattn_fused_takes(xg, 4, (blk.fn.norm.gamma,))


def test_fused_block_without_rotary_and_bias(mods):
    ops, V = mods
    blk, att, rpb = _block(V, 6)
    x = torch.randn(2, 24, 5, 9, 64)
    from oracle import unet_ref as U
    xe = x.double().permute(0, 4, 1, 2, 3)
    yn = U.channel_layernorm(xe, blk.fn.norm.gamma.detach().double())
    bb, c, f, hh, ww = yn.shape
    yn = yn.permute(0, 3, 4, 2, 1).reshape(bb, hh * ww, f, c)
    yn = U.token_attention(yn, att.to_qkv.weight.detach().double(), att.to_out.weight.detach().double(), 4, 32)
    exact = (yn.reshape(bb, hh, ww, f, c).permute(0, 4, 3, 1, 2) + xe).permute(0, 2, 3, 4, 1)
    blk = blk.to(DEV)
    with torch.no_grad():
        y = ops.temporal_attention_fused(x.to(DEV), blk.fn.norm.gamma, blk.fn.norm.eps, att.to_qkv.weight, att.to_out.weight, None, None, 4, att.scale)
    assert rel_l2(y, exact) < 1e-6


def test_unet_forward_uses_the_fused_block_when_sampling(mods):
    """The whole denoiser under no_grad (sampling): the four 64-channel temporal attentions run on csrc/attn_fused.hip and the 128-channel
    one of the second level on csrc/attn_fused_wide.hip (the 4 x 4 levels of this small input have too few sequences); result vs the oracle
    and vs the layer-by-layer path. With gradients the 64-channel blocks stay fused (one backward launch), the wide ones run layer by layer."""
    ops, V = mods
    from oracle import unet_ref as U
    torch.manual_seed(2)
    net = V.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x, t = torch.randn(1, 24, 42, 16, 16) * 0.7, torch.tensor([433])
    with torch.no_grad():
        ref = U.unet3d_forward(sd, x, t, dim=64, dim_mults=(1, 2, 4), groups=8)
        ref64 = U.unet3d_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x.double(), t, dim=64, dim_mults=(1, 2, 4), groups=8)
    net = net.to(DEV)
    try:
        with torch.no_grad():
            ops.PROFILE = {}
            out = net(x.to(DEV), t.to(DEV))
            n_fused = len(ops.PROFILE.get('tattn_fused_fwd_kernel', []))
            ops.PROFILE = None
            ops.FUSED_TATTN_WIDE = ops.FUSED_LATTN_WIDE = False      # the blocks a pass with gradients can fuse
            ops.PROFILE = {}
            out_narrow = net(x.to(DEV), t.to(DEV))
            n_narrow = len(ops.PROFILE.get('tattn_fused_fwd_kernel', []))
            ops.PROFILE = None
            ops.FUSED_TATTN = False
            out_layers = net(x.to(DEV), t.to(DEV))
    finally:
        ops.FUSED_TATTN = ops.FUSED_TATTN_WIDE = ops.FUSED_LATTN_WIDE = True
    assert n_fused == 5, n_fused          # init, downs[0], ups[2] at 16 x 16, ups[1] (64 channels at the second level) + downs[1] (128 channels, 8 x 8)
    assert n_narrow == 4, n_narrow
    e_f, e_n, e_l, e_r = rel_l2(out, ref64), rel_l2(out_narrow, ref64), rel_l2(out_layers, ref64), rel_l2(ref, ref64)
    print(f'U-Net forward: fused vs exact {e_f:.2e} (64-channel blocks only {e_n:.2e}), layers vs exact {e_l:.2e}, fp32 oracle vs exact {e_r:.2e}')
    assert rel_l2(out, ref) < 1e-5 and e_f <= 1.5 * max(e_l, e_r) + 1e-7 and e_n <= 1.5 * max(e_l, e_r) + 1e-7
    ops.PROFILE = {}
    out_g = net(x.to(DEV), t.to(DEV))               # parameters require gradients here: the 64-channel blocks still fused (the backward is one launch too)
    assert len(ops.PROFILE.get('tattn_fused_fwd_kernel', [])) == 4
    ops.PROFILE = None
    assert torch.equal(out_g.detach(), out_narrow)
    ops.FUSED_TATTN_BWD = False
    try:
        ops.PROFILE = {}
        out_g = net(x.to(DEV), t.to(DEV))           # test knob: a step with gradients runs the block layer by layer
        assert 'tattn_fused_fwd_kernel' not in ops.PROFILE
        ops.PROFILE = None
    finally:
        ops.FUSED_TATTN_BWD = True
    assert torch.equal(out_g.detach(), out_layers)


def test_unet_forward_with_48_frames_when_sampling(mods):
    """The denoiser of the super-resolution model (48 wavelet frames, inference_2d.py) under no_grad: its 64-channel temporal attentions
    run on csrc/attn_fused48.hip; result vs the fp64 oracle and the layer-by-layer path. With gradients the blocks run layer by layer."""
    ops, V = mods
    from oracle import unet_ref as U
    torch.manual_seed(3)
    net = V.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x, t = torch.randn(1, 48, 42, 16, 16) * 0.7, torch.tensor([211])
    with torch.no_grad():
        ref = U.unet3d_forward(sd, x, t, dim=64, dim_mults=(1, 2, 4), groups=8)
        ref64 = U.unet3d_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x.double(), t, dim=64, dim_mults=(1, 2, 4), groups=8)
    net = net.to(DEV)
    with torch.no_grad():
        ops.PROFILE = {}
        out = net(x.to(DEV), t.to(DEV))
        n_fused = len(ops.PROFILE.get('tattn_fused_fwd_kernel', []))
        ops.PROFILE = None
        ops.FUSED_TATTN = ops.FUSED_LATTN_WIDE = False        # (the wide linear-attention blocks too: a pass with gradients cannot take them)
        try:
            out_layers = net(x.to(DEV), t.to(DEV))
        finally:
            ops.FUSED_TATTN = ops.FUSED_LATTN_WIDE = True
    assert n_fused == 4, n_fused
    e_f, e_l, e_r = rel_l2(out, ref64), rel_l2(out_layers, ref64), rel_l2(ref, ref64)
    print(f'U-Net forward, 48 frames: fused vs exact {e_f:.2e}, layers vs exact {e_l:.2e}, fp32 oracle vs exact {e_r:.2e}')
    assert rel_l2(out, ref) < 1e-5 and e_f <= 1.5 * max(e_l, e_r) + 1e-7
    ops.PROFILE = {}
    out_g = net(x.to(DEV), t.to(DEV))               # parameters require gradients: the 48-frame block has no fused backward -> layer by layer
    assert 'tattn_fused_fwd_kernel' not in ops.PROFILE
    ops.PROFILE = None
    assert torch.equal(out_g.detach(), out_layers)


def _grads_of(blk, rpb, x, gy, dev):
    """(y, dx, {parameter: gradient}) of the block on `dev` tensors through the product path."""
    xr = x.detach().clone().to(dev).requires_grad_(True)
    for p_ in list(blk.parameters()) + list(rpb.parameters()):
        p_.grad = None
    y = blk(xr, pos_bias=rpb(24, device=dev))
    y.backward(gy.to(dev))
    g = {'gamma': blk.fn.norm.gamma.grad, 'to_qkv': blk.fn.fn.fn.to_qkv.weight.grad, 'to_out': blk.fn.fn.fn.to_out.weight.grad,
         'rel_pos': rpb.relative_attention_bias.weight.grad}
    return y.detach(), xr.grad.clone(), {k: v.clone() for k, v in g.items()}


@pytest.mark.parametrize('b,h,w', [(1, 8, 8), (3, 7, 11), (2, 40, 40)])
def test_fused_block_backward_vs_oracle_and_layers(mods, b, h, w):
    """dx, dgamma, dW_qkv, dW_out and the gradient of the relative-position embedding from ONE backward launch: against the fp64 oracle's
    autograd (exact), with the layer-by-layer HIP path and the fp32 oracle as the yardsticks; two launches give the same bits."""
    ops, V = mods
    blk, att, rpb = _block(V, 7)
    torch.manual_seed(11)
    x = torch.randn(b, 24, h, w, 64) * 1.5 + 0.2
    gy = torch.randn(b, 24, h, w, 64) * (0.5 + torch.rand(b, 1, h, w, 1) * 4.0)        # pixels with very different gradient sizes

    def oracle(dt):
        prm = {'gamma': blk.fn.norm.gamma, 'to_qkv': att.to_qkv.weight, 'to_out': att.to_out.weight, 'rel_pos': rpb.relative_attention_bias.weight}
        leaf = {k: v.detach().to(dt).requires_grad_(True) for k, v in prm.items()}
        xe = x.detach().clone().to(dt).requires_grad_(True)
        from oracle import unet_ref as U
        xc = xe.permute(0, 4, 1, 2, 3)
        bias = U.time_rel_pos_bias(leaf['rel_pos'], 24)
        y = U.channel_layernorm(xc, leaf['gamma'])
        bb, c, f, hh_, ww_ = y.shape
        y = y.permute(0, 3, 4, 2, 1).reshape(bb, hh_ * ww_, f, c)
        y = U.token_attention(y, leaf['to_qkv'], leaf['to_out'], 4, 32, freqs=att.rotary_emb.freqs.detach().to(dt), pos_bias=bias)
        y = (y.reshape(bb, hh_, ww_, f, c).permute(0, 4, 3, 1, 2) + xc).permute(0, 2, 3, 4, 1)
        y.backward(gy.to(dt))
        return xe.grad, {k: v.grad for k, v in leaf.items()}

    dx_e, g_e = oracle(torch.float64)
    dx_r, g_r = oracle(torch.float32)
    blk, rpb = blk.to(DEV), rpb.to(DEV)
    ops.PROFILE = {}
    y_f, dx_f, g_f = _grads_of(blk, rpb, x, gy, DEV)
    used = set(ops.PROFILE)
    ops.PROFILE = None
    assert 'tattn_fused_bwd_kernel' in used and not any('conv' in k or 'attn_bwd' in k or 'layernorm' in k for k in used), used
    _, dx_f2, g_f2 = _grads_of(blk, rpb, x, gy, DEV)
    assert torch.equal(dx_f, dx_f2) and all(torch.equal(g_f[k], g_f2[k]) for k in g_f)          # no atomics: bit-reproducible
    ops.FUSED_TATTN_BWD = False
    try:
        y_l, dx_l, g_l = _grads_of(blk, rpb, x, gy, DEV)
    finally:
        ops.FUSED_TATTN_BWD = True
    rec = ops._known_amax(dx_f)
    print(f'fused backward [{b},24,{h},{w},64]:')
    worst = 0.0
    for name, f_, l_, r_, e_ in [('dx', dx_f, dx_l, dx_r, dx_e)] + [(k, g_f[k], g_l[k], g_r[k], g_e[k]) for k in g_f]:
        e_f, e_l, e_r = rel_l2(f_, e_), rel_l2(l_, e_), rel_l2(r_, e_)
        print(f'  {name:8s} fused vs exact {e_f:.2e}   layer by layer vs exact {e_l:.2e}   fp32 oracle vs exact {e_r:.2e}')
        worst = max(worst, e_f / (1.5 * max(e_l, e_r) + 2e-7), e_f / 5e-6)
    assert worst <= 1.0, worst
    # the attention branch of dx alone (dx - dy), where the residual path does not mask the error
    assert rel_l2(dx_f.cpu().double() - gy.double(), dx_e - gy.double()) < 1e-5


def test_fused_backward_without_rotary_and_bias(mods):
    """The block without rotary embedding and position bias (Attention(rotary_emb=None), pos_bias=None): one backward launch against the
    fp64 oracle's autograd."""
    ops, V = mods
    torch.manual_seed(21)
    att = V.EinopsToAndFrom('b c f h w', 'b (h w) f c', V.Attention(64, heads=4, dim_head=32, rotary_emb=None))
    blk = V.Residual(V.PreNorm(64, att))
    x = torch.randn(2, 24, 6, 7, 64)
    gy = torch.randn(2, 24, 6, 7, 64)
    from oracle import unet_ref as U
    leaf = {k: v.detach().double().requires_grad_(True) for k, v in (('g', blk.fn.norm.gamma), ('q', att.fn.to_qkv.weight), ('o', att.fn.to_out.weight))}
    xe = x.double().requires_grad_(True)
    xc = xe.permute(0, 4, 1, 2, 3)
    y = U.channel_layernorm(xc, leaf['g'])
    bb, c, f, hh_, ww_ = y.shape
    y = U.token_attention(y.permute(0, 3, 4, 2, 1).reshape(bb, hh_ * ww_, f, c), leaf['q'], leaf['o'], 4, 32)
    ((y.reshape(bb, hh_, ww_, f, c).permute(0, 4, 3, 1, 2) + xc).permute(0, 2, 3, 4, 1)).backward(gy.double())
    blk = blk.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    ops.PROFILE = {}
    blk(xd).backward(gy.to(DEV))
    used, ops.PROFILE = set(ops.PROFILE), None
    assert 'tattn_fused_bwd_kernel' in used
    for name, got, want in (('dx', xd.grad, xe.grad), ('gamma', blk.fn.norm.gamma.grad, leaf['g'].grad), ('to_qkv', att.fn.to_qkv.weight.grad, leaf['q'].grad),
                            ('to_out', att.fn.to_out.weight.grad, leaf['o'].grad)):
        assert rel_l2(got, want) < 2e-6, (name, rel_l2(got, want))


def test_fused_backward_leaves_the_amax_of_dx(mods):
    ops, V = mods
    blk, att, rpb = _block(V, 8)
    blk, rpb = blk.to(DEV), rpb.to(DEV)
    x = (torch.randn(2, 24, 9, 5, 64, device=DEV) * 2).requires_grad_(True)
    seen = {}
    y = blk(x, pos_bias=rpb(24, device=DEV))
    x.register_hook(lambda g: seen.__setitem__('rec', ops._known_amax(g)))
    y.backward(torch.randn_like(y))
    assert seen.get('rec') is not None and abs(seen['rec'].max().item() - x.grad.abs().max().item()) == 0.0


def test_unet_training_gradients_with_the_fused_block(mods):
    """All parameter gradients of the denoiser with the four 64-channel temporal attention blocks fused (forward + backward launches) against
    the fp64 oracle, the layer-by-layer path as the yardstick."""
    ops, V = mods
    from oracle import unet_ref as U
    torch.manual_seed(3)
    net = V.Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=42)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x, t = torch.randn(1, 24, 42, 16, 16) * 0.7, torch.tensor([321])
    gy = torch.randn(1, 24, 42, 16, 16)
    sd64 = {k: (v.double().requires_grad_(True) if v.is_floating_point() and not k.endswith('freqs') else v) for k, v in sd.items()}
    U.unet3d_forward(sd64, x.double(), t, dim=64, dim_mults=(1, 2, 4), groups=8).backward(gy.double())
    net = net.to(DEV)

    def run():
        net.zero_grad(set_to_none=True)
        ops.PROFILE = {}
        out = net(x.to(DEV), t.to(DEV))
        out.backward(gy.to(DEV))
        prof = ops.PROFILE
        ops.PROFILE = None
        return {k: p_.grad.clone() for k, p_ in net.named_parameters() if p_.grad is not None}, prof

    g_f, prof = run()
    assert len(prof.get('tattn_fused_fwd_kernel', [])) == 4 and len(prof.get('tattn_fused_bwd_kernel', [])) == 4
    ops.FUSED_TATTN_BWD = False
    try:
        g_l, prof_l = run()
    finally:
        ops.FUSED_TATTN_BWD = True
    assert 'tattn_fused_bwd_kernel' not in prof_l
    assert set(g_f) == set(g_l)
    worst = (0.0, None)
    for k in g_f:
        e_f, e_l = rel_l2(g_f[k], sd64[k].grad), rel_l2(g_l[k], sd64[k].grad)
        assert e_f <= 1.5 * e_l + 3e-6, (k, e_f, e_l)
        if e_f > worst[0]:
            worst = (e_f, k, e_l)
    print(f'U-Net gradients with fused blocks: worst vs exact {worst[0]:.2e} ({worst[1]}; layer by layer {worst[2]:.2e})')
