"""csrc/attn_fused.hip: Residual(PreNorm(temporal attention)) of the smoke U-Net's first level (conv3d.py:165-174, 277-353) as one launch.
Checked against the oracle's restatement of the block in fp64 (the arbiter), against the layer-by-layer HIP path, and inside the whole
U-Net forward (sampling is where the fused block runs)."""
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
    rot = V.RotaryEmbedding(32)
    att = V.EinopsToAndFrom('b c f h w', 'b (h w) f c', V.Attention(64, heads=4, dim_head=32, rotary_emb=rot))
    blk = V.Residual(V.PreNorm(64, att))
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
    """The whole denoiser under no_grad (sampling): the three level-0 temporal attentions run fused; result vs the oracle and vs the
    layer-by-layer path. With gradients enabled the block is NOT fused (the training step keeps its layer-by-layer backward)."""
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
    with torch.no_grad():
        ops.PROFILE = {}
        out = net(x.to(DEV), t.to(DEV))
        n_fused = len(ops.PROFILE.get('tattn_fused_fwd_kernel', []))
        ops.PROFILE = None
        ops.FUSED_TATTN = False
        try:
            out_layers = net(x.to(DEV), t.to(DEV))
        finally:
            ops.FUSED_TATTN = True
    assert n_fused == 4, n_fused          # init, downs[0], ups[2] at 40 x 40 and ups[1] (64 channels at the second level)
    e_f, e_l, e_r = rel_l2(out, ref64), rel_l2(out_layers, ref64), rel_l2(ref, ref64)
    print(f'U-Net forward: fused vs exact {e_f:.2e}, layers vs exact {e_l:.2e}, fp32 oracle vs exact {e_r:.2e}')
    assert rel_l2(out, ref) < 1e-5 and e_f <= 1.5 * max(e_l, e_r) + 1e-7
    ops.PROFILE = {}
    out_g = net(x.to(DEV), t.to(DEV))               # parameters require gradients here
    assert 'tattn_fused_fwd_kernel' not in ops.PROFILE
    ops.PROFILE = None
    assert torch.equal(out_g.detach(), out_layers)
