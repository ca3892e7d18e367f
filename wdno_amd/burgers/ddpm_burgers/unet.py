"""Unet2D on MI355X -- drop-in for burgers/ddpm_burgers/unet.py:263-411 (the (t, x) plane treated as an image).

Same class name, constructor signature, attributes (.channels, .self_condition, .out_dim) and state_dict keys/shapes
as the reference, so train_ddpm_burgers.py:149-156 builds it unchanged and upstream checkpoints load. Internally the
activations are channels-last [B, H, W, C] end to end and every operator is a HIP launch (wdno_amd.ops):

  * Downsample2d = pixel-unshuffle + 1x1 conv in the reference (unet.py:41-45) is ONE 2x2 / stride-2 convolution here:
    the 1x1 weight [C', 4C, 1, 1] read as [C', C, 2, 2] is exactly that kernel, so the shuffled tensor never exists;
  * GroupNorm + (scale+1, shift) + SiLU is one fused normalisation (groups = 1 -> whole-sample reduction);
  * residual adds ride in convolution epilogues where a convolution ends the branch.

Unet1D / RMSNorm / 1-D variants of the reference file are dead code on the WDNO path and are not provided.
"""
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


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        if isinstance(self.fn, PreNorm):          # x reaches the residual add THROUGH the norm (ops.layernorm_cl_skip)
            return self.fn(x, residual=True)
        return self.fn(x, residual=x)


class LayerNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))

    def forward(self, x, out_planes=False):
        return ops.layernorm_cl(x, self.g, 1e-5, out_planes)


class PreNorm(nn.Module):
    def __init__(self, dim, fn, conv_2d=False):
        super().__init__()
        if not conv_2d:
            raise NotImplementedError('only the 2-D (conv_2d=True) path exists on the WDNO hot path')
        self.fn = fn
        self.norm = LayerNorm(dim)

    def forward(self, x, residual=None):
        # the normed tensor is read by fn's to_qkv projection only: where that one takes fp16 planes, the norm writes them
        planes = hasattr(self.fn, 'to_qkv') and ops.conv_reads_planes(x.numel() // x.shape[-1], self.fn.to_qkv.weight)
        if residual is True:
            y, xs = ops.layernorm_cl_skip(x, self.norm.g, 1e-5, planes)
            return self.fn(y, residual=xs)
        return self.fn(self.norm(x, planes), residual=residual)


def Upsample2d(dim, dim_out=None):
    return nn.Sequential(nn.Upsample(scale_factor=2, mode='nearest'), nn.Conv2d(dim, default(dim_out, dim), 3, padding=1))


def Downsample2d(dim, dim_out=None):
    # index 0 stands for the reference's einops Rearrange (no parameters); see module docstring
    return nn.Sequential(nn.Identity(), nn.Conv2d(dim * 4, default(dim_out, dim), 1))


class Block(nn.Module):
    def __init__(self, dim, dim_out, groups=8, conv_2d=False):
        super().__init__()
        assert conv_2d
        self.proj = nn.Conv2d(dim, dim_out, 3, padding=1)
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
    def __init__(self, dim, dim_out, *, time_emb_dim=None, groups=8, conv_2d=False):
        super().__init__()
        assert conv_2d
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, dim_out * 2)) if exists(time_emb_dim) else None
        self.block1 = Block(dim, dim_out, groups=groups, conv_2d=True)
        self.block2 = Block(dim_out, dim_out, groups=groups, conv_2d=True)
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()
        self.conv_2d = conv_2d

    def forward(self, x, time_emb=None, scale_shift=None):
        """scale_shift: this block's projection of the time embedding when the caller computed all blocks' projections in one launch
        (Unet2D.time_projections); otherwise it is computed here."""
        if scale_shift is None and exists(self.mlp) and exists(time_emb):
            scale_shift = ops.conv_cl(ops.silu_shared(time_emb), self.mlp[1].weight, self.mlp[1].bias)   # [B, 2C] = (scale | shift)
        # block1's output is read by block2's convolution only: where that one takes fp16 planes, the norm writes them
        planes = ops.conv_reads_planes(x.numel() // x.shape[-1], self.block2.proj.weight)
        h, xs = self.block1(x, scale_shift=scale_shift, with_skip=True, out_planes=planes)
        if isinstance(self.res_conv, nn.Identity):
            return self.block2(h, residual=xs)
        h = self.block2(h)
        return ops.conv_cl(xs, self.res_conv.weight, self.res_conv.bias, residual=h)


class LinearAttention(nn.Module):
    def __init__(self, dim, heads=4, dim_head=32, conv_2d=False):
        super().__init__()
        assert conv_2d and dim_head == 32
        self.scale = dim_head ** -0.5
        self.heads = heads
        hidden_dim = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden_dim * 3, 1, bias=False)
        self.to_out = nn.Sequential(nn.Conv2d(hidden_dim, dim, 1), LayerNorm(dim))
        self.conv_2d = conv_2d

    def forward(self, x, residual=None):
        b, h, w, _ = x.shape
        qkv = ops.conv_cl(x, self.to_qkv.weight, grad_planes=True)         # read by the attention kernels only
        planes = ops.conv_reads_planes(b * h * w, self.to_out[0].weight)              # out is read by to_out only
        out = ops.linear_attention(qkv, b, h * w, self.heads, self.scale, out_planes=planes)
        out = ops.conv_cl(out, self.to_out[0].weight, self.to_out[0].bias)
        out = self.to_out[1](out)
        return out if residual is None else ops.add(out, residual)


class Attention(nn.Module):
    def __init__(self, dim, heads=4, dim_head=32, conv_2d=False):
        super().__init__()
        assert conv_2d and dim_head == 32
        self.scale = dim_head ** -0.5
        self.heads = heads
        hidden_dim = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden_dim * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hidden_dim, dim, 1)
        self.conv_2d = conv_2d

    def forward(self, x, residual=None):
        b, h, w, _ = x.shape
        qkv = ops.conv_cl(x, self.to_qkv.weight, grad_planes=True)         # read by the attention kernels only
        out = ops.softmax_attention(qkv, self.heads, b, 1, h * w, h * w, 0, 1, self.scale)
        return ops.conv_cl(out, self.to_out.weight, self.to_out.bias, residual=residual)


class Unet2D(nn.Module):
    def __init__(
        self,
        dim,
        init_dim=None,
        out_dim=None,
        dim_mults=(1, 2, 4, 8),
        channels=2,
        self_condition=False,
        resnet_block_groups=8,
        learned_variance=False,
        learned_sinusoidal_cond=False,
        random_fourier_features=False,
        learned_sinusoidal_dim=16,
        sinusoidal_pos_emb_theta=10000,
        attn_dim_head=32,
        attn_heads=4,
    ):
        super().__init__()
        if learned_sinusoidal_cond or random_fourier_features:
            raise NotImplementedError('learned / random Fourier time embeddings are never enabled on the WDNO path')
        self.channels = channels
        self.self_condition = self_condition
        input_channels = channels * (2 if self_condition else 1)
        self.dim = dim
        self.theta = sinusoidal_pos_emb_theta
        self.random_or_learned_sinusoidal_cond = False

        time_dim = dim * 4
        self.time_mlp = nn.Sequential(nn.Identity(), nn.Linear(dim, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        block_klass = partial(ResnetBlock, groups=resnet_block_groups, conv_2d=True, time_emb_dim=time_dim)

        init_dim = default(init_dim, dim)
        self.init_conv = nn.Conv2d(input_channels, init_dim, 7, padding=3)
        dims = [init_dim, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))

        self.downs = nn.ModuleList([])
        num_resolutions = len(in_out)
        for ind, (dim_in, dim_out) in enumerate(in_out):
            is_last = ind >= (num_resolutions - 1)
            self.downs.append(nn.ModuleList([
                block_klass(dim_in, dim_in),
                block_klass(dim_in, dim_in),
                Residual(PreNorm(dim_in, LinearAttention(dim_in, conv_2d=True), conv_2d=True)),
                Downsample2d(dim_in, dim_out) if not is_last else nn.Conv2d(dim_in, dim_out, 3, padding=1),
            ]))

        mid_dim = dims[-1]
        self.mid_block1 = block_klass(mid_dim, mid_dim)
        self.mid_attn = Residual(PreNorm(mid_dim, Attention(mid_dim, conv_2d=True, dim_head=attn_dim_head, heads=attn_heads), conv_2d=True))
        self.mid_block2 = block_klass(mid_dim, mid_dim)

        self.ups = nn.ModuleList([])
        for ind, (dim_in, dim_out) in enumerate(reversed(in_out)):
            is_last = ind == (len(in_out) - 1)
            self.ups.append(nn.ModuleList([
                block_klass(dim_out + dim_in, dim_out),
                block_klass(dim_out + dim_in, dim_out),
                Residual(PreNorm(dim_out, LinearAttention(dim_out, conv_2d=True), conv_2d=True)),
                Upsample2d(dim_out, dim_in) if not is_last else nn.Conv2d(dim_out, dim_in, 3, padding=1),
            ]))

        default_out_dim = channels * (1 if not learned_variance else 2)
        self.out_dim = default(out_dim, default_out_dim)
        self.final_res_block = block_klass(dim * 2, dim)
        self.final_conv = nn.Conv2d(dim, self.out_dim, 1)

    def time_embedding(self, time):
        e = ops.sinusoidal_embedding(time, self.dim, self.theta)
        e = ops.conv_cl(e, self.time_mlp[1].weight, self.time_mlp[1].bias)
        e = ops.gelu(e)
        return ops.conv_cl(e, self.time_mlp[3].weight, self.time_mlp[3].bias)

    def time_projections(self, t):
        """The scale/shift projections of all ResnetBlocks, in the order forward() runs them, from one grouped launch (ops.linear_multi: they
        all read silu(t)); an iterator of Nones when the grouped kernels do not take the shapes (each block then projects for itself)."""
        blocks = getattr(self, '_time_blocks', None)
        if blocks is None:
            blocks = ([b for lv in self.downs for b in lv[:2]] + [self.mid_block1, self.mid_block2] + [b for lv in self.ups for b in lv[:2]]
                      + [self.final_res_block])
            object.__setattr__(self, '_time_blocks', blocks)         # (not a submodule list: the blocks are registered where the reference has them)
        out = ops.linear_multi(ops.silu_shared(t), [b.mlp[1] for b in blocks]) if all(exists(b.mlp) for b in blocks) else None
        return iter(out if out is not None else [None] * len(blocks))

    def forward(self, x, time, x_self_cond=None):
        """x: [B, C, H, W] -> [B, out_dim, H, W]"""
        if self.self_condition:
            x_self_cond = default(x_self_cond, lambda: torch.zeros_like(x))
            x = torch.cat((x_self_cond, x), dim=1)
        x = ops.nc_to_cl(x)
        x = ops.conv_cl(x, self.init_conv.weight, self.init_conv.bias, padding=3)
        r = x
        t = self.time_embedding(time)
        ss = self.time_projections(t)
        hs = []
        for block1, block2, attn, downsample in self.downs:
            x = block1(x, t, next(ss))
            hs.append(x)
            x = block2(x, t, next(ss))
            x = attn(x)
            hs.append(x)
            if isinstance(downsample, nn.Sequential):
                wgt = downsample[1].weight                       # [C', 4C, 1, 1] == [C', C, 2, 2] of a 2x2 / s2 convolution
                x = ops.conv_cl(x, wgt.view(wgt.shape[0], wgt.shape[1] // 4, 2, 2), downsample[1].bias, stride=2, padding=0)
            else:
                x = ops.conv_cl(x, downsample.weight, downsample.bias, padding=1)
        x = self.mid_block1(x, t, next(ss))
        x = self.mid_attn(x)
        x = self.mid_block2(x, t, next(ss))
        for block1, block2, attn, upsample in self.ups:
            npx = x.numel() // x.shape[-1]
            x = block1(ops.concat_cl(x, hs.pop(), planes_only=ops.resnet_reads_planes(npx, block1)), t, next(ss))
            x = block2(ops.concat_cl(x, hs.pop(), planes_only=ops.resnet_reads_planes(npx, block2)), t, next(ss))
            x = attn(x)
            if isinstance(upsample, nn.Sequential):
                x = ops.conv_cl(ops.upsample2x_cl(x), upsample[1].weight, upsample[1].bias, padding=1)
            else:
                x = ops.conv_cl(x, upsample.weight, upsample.bias, padding=1)
        x = self.final_res_block(ops.concat_cl(x, r, planes_only=ops.resnet_reads_planes(x.numel() // x.shape[-1], self.final_res_block)), t, next(ss))
        x = ops.conv_cl(x, self.final_conv.weight, self.final_conv.bias)
        return ops.cl_to_nc(x, self.out_dim)
