"""Unet3D_with_Conv3D on MI355X -- drop-in for smoke/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:357-574.

Same class name, constructor signature, attributes and state_dict keys/shapes as the reference (checked against
tests/golden/ref_manifest.json), so upstream checkpoints load and smoke/train_2d.py:94-98 constructs it unchanged.
The computation is re-designed: activations stay channels-last [B, F, H, W, C] from the first convolution to the
last, every operator is a launch of libwdno_hip.so (wdno_amd.ops), residual adds and biases ride in the convolution
epilogues, and the einops round trips of the reference (b c f h w <-> b (h w) f c) disappear because the attention
kernels address tokens by stride.

torch.nn layers (Conv3d, Linear, GroupNorm, Embedding) are used as parameter containers only -- their forward() is
never called -- which keeps initialisation and state_dict naming identical to the reference.
"""
import math
from functools import partial

import torch
from torch import nn

from wdno_amd import ops


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def is_odd(n):
    return (n % 2) == 1


class RotaryEmbedding(nn.Module):
    """Holds the `freqs` tensor of rotary_embedding_torch.RotaryEmbedding (a frozen parameter in the state_dict)."""

    def __init__(self, dim, theta=10000):
        super().__init__()
        self.freqs = nn.Parameter(1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)), requires_grad=False)


class RelativePositionBias(nn.Module):
    """T5-style bucketed bias (conv3d.py:74-112). The bucket table is integer work done once on the host."""

    def __init__(self, heads=8, num_buckets=32, max_distance=128):
        super().__init__()
        self.num_buckets = num_buckets
        self.max_distance = max_distance
        self.relative_attention_bias = nn.Embedding(num_buckets, heads)
        self._buckets = {}

    def bucket_table(self, n, device):
        key = (n, str(device))
        if key not in self._buckets:
            nb = self.num_buckets // 2
            max_exact = nb // 2
            pos = torch.arange(n, dtype=torch.long)
            rel = pos[None, :] - pos[:, None]
            m = -rel
            ret = (m < 0).long() * nb
            m = m.abs()
            large = max_exact + (torch.log(m.float() / max_exact) / math.log(self.max_distance / max_exact) * (nb - max_exact)).long()
            large = torch.minimum(large, torch.full_like(large, nb - 1))
            self._buckets[key] = (ret + torch.where(m < max_exact, m, large)).to(device)
        return self._buckets[key]

    def forward(self, n, device):
        return ops.relpos_bias(self.relative_attention_bias.weight, self.bucket_table(n, device))   # [heads, n, n]


class LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(1, dim, 1, 1, 1))

    def forward(self, x, out_planes=False):
        return ops.layernorm_cl(x, self.gamma, self.eps, out_planes)


class Residual(nn.Module):
    """fn(x) + x. The add is folded into the last convolution of `fn` (passed down as `residual`)."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, **kwargs):
        if isinstance(self.fn, PreNorm):          # x reaches the residual add THROUGH the norm (ops.layernorm_cl_skip)
            return self.fn(x, residual=True, **kwargs)
        return self.fn(x, residual=x, **kwargs)


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = LayerNorm(dim)

    def forward(self, x, residual=None, **kwargs):
        # the normed tensor is read by fn's to_qkv projection only: where that one takes fp16 planes, the norm writes them
        att = self.fn if hasattr(self.fn, 'to_qkv') else getattr(self.fn, 'fn', None)
        if (residual is True and getattr(self.fn, 'token_axis', None) == 'frames' and kwargs.get('focus_present_mask') is None
                and ops.tattn_fused_takes(x, att.heads, (self.norm.gamma, att.to_qkv.weight, att.to_out.weight, kwargs.get('pos_bias')))):
            # norm -> to_qkv -> rotary / attention over the frames -> to_out -> + x as one launch (csrc/attn_fused.hip)
            rot = ops.rotary_tables(att.rotary_emb.freqs, x.shape[1]) if exists(att.rotary_emb) else None
            return ops.temporal_attention_fused(x, self.norm.gamma, self.norm.eps, att.to_qkv.weight, att.to_out.weight, rot,
                                                kwargs.get('pos_bias'), att.heads, att.scale)
        if (residual is True and isinstance(self.fn, SpatialLinearAttention)
                and ops.lattn_fused_takes(x, att.heads, (self.norm.gamma, att.to_qkv.weight, att.to_out.weight, att.to_out.bias))):
            # norm -> to_qkv -> linear attention -> to_out -> + x without the [pixels x 384] projections (csrc/linattn_fused.hip; no gradient)
            return ops.linear_attention_fused(x, self.norm.gamma, self.norm.eps, att.to_qkv.weight, att.to_out.weight, att.to_out.bias,
                                              att.heads, att.scale)
        planes = hasattr(att, 'to_qkv') and ops.conv_reads_planes(x.numel() // x.shape[-1], att.to_qkv.weight)
        if residual is True:
            y, xs = ops.layernorm_cl_skip(x, self.norm.gamma, self.norm.eps, planes)
            return self.fn(y, residual=xs, **kwargs)
        return self.fn(self.norm(x, planes), residual=residual, **kwargs)


class Block(nn.Module):
    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.proj = nn.Conv3d(dim, dim_out, (3, 3, 3), padding=(1, 1, 1))
        self.norm = nn.GroupNorm(groups, dim_out)
        self.act = nn.SiLU()
        self.groups = groups

    def forward(self, x, scale_shift=None, with_skip=False, out_planes=False, residual=None):
        if with_skip:              # block input that also feeds the skip connection: handed through the convolution (ops.conv_cl_skip)
            x, xs = ops.conv_cl_skip(x, self.proj.weight, self.proj.bias, padding=1, grad_planes=True, to_norm=True)
            return ops.groupnorm_act(x, self.norm.weight, self.norm.bias, self.groups, scale_shift, act=True, eps=self.norm.eps, out_planes=out_planes), xs
        x = ops.conv_cl(x, self.proj.weight, self.proj.bias, padding=1, grad_planes=True, to_norm=True)     # x goes to the norm and nowhere else
        if residual is not None:   # identity skip of the ResnetBlock: added in the norm's apply pass
            return ops.groupnorm_act_add(x, self.norm.weight, self.norm.bias, self.groups, residual, scale_shift, act=True, eps=self.norm.eps)
        return ops.groupnorm_act(x, self.norm.weight, self.norm.bias, self.groups, scale_shift, act=True, eps=self.norm.eps, out_planes=out_planes)


class ResnetBlock(nn.Module):
    def __init__(self, dim, dim_out, *, time_emb_dim=None, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, dim_out * 2)) if exists(time_emb_dim) else None
        self.block1 = Block(dim, dim_out, groups=groups)
        self.block2 = Block(dim_out, dim_out, groups=groups)
        self.res_conv = nn.Conv3d(dim, dim_out, 1) if dim != dim_out else nn.Identity()

    def forward(self, x, time_emb=None, scale_shift=None):
        """scale_shift: this block's projection of the time embedding when the caller has computed the projections of all blocks in one
        launch (Unet3D_with_Conv3D.time_projections); otherwise it is computed here."""
        if exists(self.mlp) and scale_shift is None:
            assert exists(time_emb), 'time emb must be passed in'
            # [B, 2*C]: first half = scale, second half = shift (chunk(2, dim=1) in the reference)
            scale_shift = ops.conv_cl(ops.silu_shared(time_emb), self.mlp[1].weight, self.mlp[1].bias)
        # block1's output is read by block2's convolution only: where that one takes fp16 planes, the norm writes them
        planes = ops.conv_reads_planes(x.numel() // x.shape[-1], self.block2.proj.weight)
        h, xs = self.block1(x, scale_shift=scale_shift, with_skip=True, out_planes=planes)
        if isinstance(self.res_conv, nn.Identity):
            return self.block2(h, residual=xs)
        h = self.block2(h)
        return ops.conv_cl(xs, self.res_conv.weight, self.res_conv.bias, residual=h)


class SpatialLinearAttention(nn.Module):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        assert dim_head == 32, 'the HIP attention kernels are specialised for dim_head = 32 (the reference default)'
        self.scale = dim_head ** -0.5
        self.heads = heads
        hidden_dim = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden_dim * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hidden_dim, dim, 1)

    def forward(self, x, residual=None):
        b, f, h, w, _ = x.shape
        qkv = ops.conv_cl(x, self.to_qkv.weight, grad_planes=True)         # read by the attention kernels only
        planes = ops.conv_reads_planes(b * f * h * w, self.to_out.weight)             # out is read by to_out only
        out = ops.linear_attention(qkv, b * f, h * w, self.heads, self.scale, out_planes=planes)      # rows in CL order; same leading shape out
        return ops.conv_cl(out, self.to_out.weight, self.to_out.bias, residual=residual)


class Attention(nn.Module):
    def __init__(self, dim, heads=4, dim_head=32, rotary_emb=None):
        super().__init__()
        assert dim_head == 32, 'the HIP attention kernels are specialised for dim_head = 32 (the reference default)'
        self.scale = dim_head ** -0.5
        self.heads = heads
        hidden_dim = dim_head * heads
        self.rotary_emb = rotary_emb
        self.to_qkv = nn.Linear(dim, hidden_dim * 3, bias=False)
        self.to_out = nn.Linear(hidden_dim, dim, bias=False)


class EinopsToAndFrom(nn.Module):
    """In the reference this rearranges 'b c f h w' into token-major form around `fn`. Here the tensor never moves:
    the pattern only selects which axis the attention kernel treats as the token axis."""

    def __init__(self, from_einops, to_einops, fn):
        super().__init__()
        self.from_einops = from_einops
        self.to_einops = to_einops
        self.fn = fn
        if to_einops == 'b (h w) f c':
            self.token_axis = 'frames'
        elif to_einops == 'b f (h w) c':
            self.token_axis = 'pixels'
        else:
            raise ValueError(f'unsupported attention layout {to_einops!r}')

    def forward(self, x, residual=None, pos_bias=None, focus_present_mask=None):
        if focus_present_mask is not None and bool(focus_present_mask.any()):
            raise NotImplementedError('focus_present_mask is always all-False on the WDNO path (conv3d.py:304,332)')
        att = self.fn
        b, f, h, w, _ = x.shape
        rows = ops.conv_cl(x, att.to_qkv.weight, grad_planes=True)     # [b, f, h, w, 3*hidden]: the attention kernels index its rows in place; read by them only
        planes = ops.conv_reads_planes(b * f * h * w, att.to_out.weight)              # out is read by to_out (and the attention backward) only
        if self.token_axis == 'frames':
            rot = ops.rotary_tables(att.rotary_emb.freqs, f) if exists(att.rotary_emb) else None
            out = ops.softmax_attention(rows, att.heads, b, h * w, f, f * h * w, 1, h * w, att.scale, bias=pos_bias, rot=rot, out_planes=planes)
        else:
            out = ops.softmax_attention(rows, att.heads, b * f, 1, h * w, h * w, 0, 1, att.scale, bias=pos_bias, rot=None, out_planes=planes)
        return ops.conv_cl(out, att.to_out.weight, None, residual=residual)


def Upsample(dim):
    return nn.ConvTranspose3d(dim, dim, (1, 4, 4), (1, 2, 2), (0, 1, 1))


def Downsample(dim):
    return nn.Conv3d(dim, dim, (1, 4, 4), (1, 2, 2), (0, 1, 1))


class Unet3D_with_Conv3D(nn.Module):
    def __init__(
        self,
        dim,
        cond_dim=None,
        out_dim=None,
        dim_mults=(1, 2, 4, 8),
        channels=6,
        attn_heads=4,
        attn_dim_head=32,
        use_bert_text_cond=False,
        init_dim=None,
        init_kernel_size=7,
        use_sparse_linear_attn=True,
        block_type='resnet',
        resnet_groups=8,
    ):
        super().__init__()
        if exists(cond_dim) or use_bert_text_cond:
            raise NotImplementedError('text / vector conditioning is never enabled on the WDNO path (train_2d.py:94-98)')
        self.channels = channels
        self.self_condition = False
        self.has_cond = False
        self.null_cond_emb = None
        self.dim = dim

        rotary_emb = RotaryEmbedding(min(32, attn_dim_head))
        temporal_attn = lambda d: EinopsToAndFrom('b c f h w', 'b (h w) f c',
                                                  Attention(d, heads=attn_heads, dim_head=attn_dim_head, rotary_emb=rotary_emb))
        self.time_rel_pos_bias = RelativePositionBias(heads=attn_heads, max_distance=32)

        init_dim = default(init_dim, dim)
        assert is_odd(init_kernel_size)
        init_padding = init_kernel_size // 2
        self.init_conv = nn.Conv3d(channels, init_dim, (init_kernel_size,) * 3, padding=(init_padding,) * 3)
        self.init_temporal_attn = Residual(PreNorm(init_dim, temporal_attn(init_dim)))

        dims = [init_dim, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))

        time_dim = dim * 4
        # index 0 keeps the reference's Sequential numbering (SinusoidalPosEmb has no parameters)
        self.time_mlp = nn.Sequential(nn.Identity(), nn.Linear(dim, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        num_resolutions = len(in_out)
        block_klass = partial(ResnetBlock, groups=resnet_groups)
        block_klass_cond = partial(block_klass, time_emb_dim=time_dim)

        for ind, (dim_in, dim_out) in enumerate(in_out):
            is_last = ind >= (num_resolutions - 1)
            self.downs.append(nn.ModuleList([
                block_klass_cond(dim_in, dim_out),
                block_klass_cond(dim_out, dim_out),
                Residual(PreNorm(dim_out, SpatialLinearAttention(dim_out, heads=attn_heads))) if use_sparse_linear_attn else nn.Identity(),
                Residual(PreNorm(dim_out, temporal_attn(dim_out))),
                Downsample(dim_out) if not is_last else nn.Identity(),
            ]))

        mid_dim = dims[-1]
        self.mid_block1 = block_klass_cond(mid_dim, mid_dim)
        spatial_attn = EinopsToAndFrom('b c f h w', 'b f (h w) c', Attention(mid_dim, heads=attn_heads))
        self.mid_spatial_attn = Residual(PreNorm(mid_dim, spatial_attn))
        self.mid_temporal_attn = Residual(PreNorm(mid_dim, temporal_attn(mid_dim)))
        self.mid_block2 = block_klass_cond(mid_dim, mid_dim)

        for ind, (dim_in, dim_out) in enumerate(reversed(in_out)):
            is_last = ind >= (num_resolutions - 1)
            self.ups.append(nn.ModuleList([
                block_klass_cond(dim_out * 2, dim_in),
                block_klass_cond(dim_in, dim_in),
                Residual(PreNorm(dim_in, SpatialLinearAttention(dim_in, heads=attn_heads))) if use_sparse_linear_attn else nn.Identity(),
                Residual(PreNorm(dim_in, temporal_attn(dim_in))),
                Upsample(dim_in) if not is_last else nn.Identity(),
            ]))

        out_dim = default(out_dim, channels)
        self.out_dim = out_dim
        self.final_conv = nn.Sequential(block_klass(dim * 2, dim), nn.Conv3d(dim, out_dim, 1))

    def forward_with_cond_scale(self, *args, cond_scale=2., **kwargs):
        return self.forward(*args, null_cond_prob=0., **kwargs)      # has_cond is always False (conv3d.py:481)

    def time_embedding(self, time):
        e = ops.sinusoidal_embedding(time, self.dim)
        e = ops.conv_cl(e, self.time_mlp[1].weight, self.time_mlp[1].bias)
        e = ops.gelu(e)
        return ops.conv_cl(e, self.time_mlp[3].weight, self.time_mlp[3].bias)

    def time_projections(self, t):
        """The scale/shift projections of all ResnetBlocks, in the order forward() runs them, from one grouped launch (ops.linear_multi: they
        all read silu(t)); an iterator of Nones when the grouped kernels do not take the shapes (each block then projects for itself)."""
        blocks = getattr(self, '_time_blocks', None)
        if blocks is None:
            blocks = [b for lv in self.downs for b in lv[:2]] + [self.mid_block1, self.mid_block2] + [b for lv in self.ups for b in lv[:2]]
            object.__setattr__(self, '_time_blocks', blocks)         # (not a submodule list: the blocks are registered where the reference has them)
        out = ops.linear_multi(ops.silu_shared(t), [b.mlp[1] for b in blocks]) if all(exists(b.mlp) for b in blocks) else None
        return iter(out if out is not None else [None] * len(blocks))

    def forward(self, x, time, cond=None, null_cond_prob=0., focus_present_mask=None, prob_focus_present=0.):
        """x: [B, F, C, H, W] (frames before channels, as handed over by GaussianDiffusion) -> same shape."""
        assert cond is None, 'cond must be None (has_cond is False on the WDNO path)'
        if prob_focus_present != 0.:
            raise NotImplementedError('prob_focus_present is always 0 on the WDNO path')
        b, f, c, h, w = x.shape
        # [B*F, C, H, W] -> channels-last [B, F, H, W, Cp]
        x_api = x
        x = ops.nc_to_cl(x.reshape(b * f, c, h, w)).reshape(b, f, h, w, -1)
        ops.carry_zero_box(x, x_api)       # the sampler / p_losses left structural zeros (pad condition): the stem skips the stages that only see them
        pos_bias = self.time_rel_pos_bias(f, device=x.device)

        x = ops.conv_cl(x, self.init_conv.weight, self.init_conv.bias, padding=self.init_conv.padding)
        x = self.init_temporal_attn(x, pos_bias=pos_bias)
        r = x
        t = self.time_embedding(time)
        ss = self.time_projections(t)

        hs = []
        for block1, block2, spatial_attn, temporal_attn, downsample in self.downs:
            x = block1(x, t, next(ss))
            x = block2(x, t, next(ss))
            if not isinstance(spatial_attn, nn.Identity):
                x = spatial_attn(x)
            x = temporal_attn(x, pos_bias=pos_bias, focus_present_mask=focus_present_mask)
            hs.append(x)
            if not isinstance(downsample, nn.Identity):
                x = ops.conv_cl(x, downsample.weight, downsample.bias, stride=(1, 2, 2), padding=(0, 1, 1))

        x = self.mid_block1(x, t, next(ss))
        x = self.mid_spatial_attn(x)
        x = self.mid_temporal_attn(x, pos_bias=pos_bias, focus_present_mask=focus_present_mask)
        x = self.mid_block2(x, t, next(ss))

        for block1, block2, spatial_attn, temporal_attn, upsample in self.ups:
            x = ops.concat_cl(x, hs.pop(), planes_only=ops.resnet_reads_planes(x.numel() // x.shape[-1], block1))
            x = block1(x, t, next(ss))
            x = block2(x, t, next(ss))
            if not isinstance(spatial_attn, nn.Identity):
                x = spatial_attn(x)
            x = temporal_attn(x, pos_bias=pos_bias, focus_present_mask=focus_present_mask)
            if not isinstance(upsample, nn.Identity):
                x = ops.conv_transpose_cl(x, upsample.weight, upsample.bias)

        x = ops.concat_cl(x, r, planes_only=ops.resnet_reads_planes(x.numel() // x.shape[-1], self.final_conv[0]))
        x = self.final_conv[0](x)
        x = ops.conv_cl(x, self.final_conv[1].weight, self.final_conv[1].bias)
        bb, ff, hh, ww, kp = x.shape
        return ops.cl_to_nc(x.reshape(bb * ff, hh, ww, kp), self.out_dim).reshape(bb, ff, self.out_dim, hh, ww)
