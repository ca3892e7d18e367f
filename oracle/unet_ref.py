"""ORACLE (test infrastructure only) -- functional CPU restatement of the two WDNO denoiser U-Nets.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Plain torch fp32 ops on NCHW / NCDHW tensors, weights passed as a flat {state_dict key: tensor} mapping so
that the same dictionary can be loaded into the HIP modules. Autograd through these functions provides the
reference gradients. Pinned against tests/golden/ref_unet{2d,3d}_*.npz (outputs of the reference itself,
made by tests/golden/make_ref_golden.py); see tests/test_oracle_unet.py.

Follows
  burgers/ddpm_burgers/unet.py:18-45,55-108,129-259,263-411          (Unet2D and its blocks)
  smoke/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py:74-112,131-184,189-353,357-574
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- shared pieces
def sinusoidal_embedding(t, dim, theta=10000.0):
    """unet.py:88-96 / conv3d.py:144-151 : cat(sin(t f_k), cos(t f_k)), f_k = exp(-k ln(theta)/(half-1))."""
    half = dim // 2
    f = torch.exp(torch.arange(half, device=t.device) * -(math.log(theta) / (half - 1)))
    e = t[:, None] * f[None, :]
    return torch.cat([e.sin(), e.cos()], dim=-1)


def time_mlp(sd, pfx, t, dim):
    """[SinusoidalPosEmb, Linear, GELU(erf), Linear]   unet.py:301-306 / conv3d.py:405-410."""
    w1 = sd[pfx + '1.weight']
    e = sinusoidal_embedding(t, dim).to(w1.dtype)      # always built in fp32 like the reference; upcast only for an fp64 evaluation
    e = F.linear(e, w1, sd[pfx + '1.bias'])
    e = F.gelu(e)
    return F.linear(e, sd[pfx + '3.weight'], sd[pfx + '3.bias'])


def _conv(x, w, b=None, **kw):
    return (F.conv2d if w.dim() == 4 else F.conv3d)(x, w, b, **kw)


def resnet_block(sd, pfx, x, temb, groups):
    """unet.py:150-181 / conv3d.py:206-230. block1 gets (scale+1, shift) from the time MLP; block2 does not."""
    nd = x.dim() - 2
    pad = 1
    h = _conv(x, sd[pfx + 'block1.proj.weight'], sd[pfx + 'block1.proj.bias'], padding=pad)
    h = F.group_norm(h, groups, sd[pfx + 'block1.norm.weight'], sd[pfx + 'block1.norm.bias'], eps=1e-5)
    if temb is not None and (pfx + 'mlp.1.weight') in sd:
        e = F.linear(F.silu(temb), sd[pfx + 'mlp.1.weight'], sd[pfx + 'mlp.1.bias'])
        e = e.reshape(*e.shape, *([1] * nd))
        scale, shift = e.chunk(2, dim=1)
        h = h * (scale + 1) + shift
    h = F.silu(h)
    h = _conv(h, sd[pfx + 'block2.proj.weight'], sd[pfx + 'block2.proj.bias'], padding=pad)
    h = F.group_norm(h, groups, sd[pfx + 'block2.norm.weight'], sd[pfx + 'block2.norm.bias'], eps=1e-5)
    h = F.silu(h)
    if (pfx + 'res_conv.weight') in sd:
        x = _conv(x, sd[pfx + 'res_conv.weight'], sd[pfx + 'res_conv.bias'])
    return h + x


def channel_layernorm(x, g, eps=1e-5):
    """unet.py:55-65 / conv3d.py:165-174 : biased variance over dim 1, gain only."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) / (var + eps).sqrt() * g


def linear_attention_2d(x, w_qkv, w_out, b_out, heads=4, dim_head=32):
    """unet.py:203-223 / conv3d.py:241-258 on [B, C, H, W]; returns to_out[0] output (before any trailing norm)."""
    b, c, hh, ww = x.shape
    q, k, v = F.conv2d(x, w_qkv).chunk(3, dim=1)
    q, k, v = (z.reshape(b, heads, dim_head, hh * ww) for z in (q, k, v))
    q = q.softmax(dim=-2) * dim_head ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum('bhdn,bhen->bhde', k, v)
    out = torch.einsum('bhde,bhdn->bhen', ctx, q).reshape(b, heads * dim_head, hh, ww)
    return F.conv2d(out, w_out, b_out)


# ----------------------------------------------------------------------------- Burgers Unet2D
def unet2d_forward(sd, x, t, *, dim, dim_mults=(1, 2, 4, 8), groups=1, heads=4, dim_head=32):
    """burgers/ddpm_burgers/unet.py:372-411 (self_condition=False)."""
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n_res = len(in_out)

    def lin_attn(pfx, x):
        y = channel_layernorm(x, sd[pfx + 'fn.norm.g'])
        y = linear_attention_2d(y, sd[pfx + 'fn.fn.to_qkv.weight'], sd[pfx + 'fn.fn.to_out.0.weight'],
                                sd[pfx + 'fn.fn.to_out.0.bias'], heads, dim_head)
        y = channel_layernorm(y, sd[pfx + 'fn.fn.to_out.1.g'])
        return y + x

    x = F.conv2d(x, sd['init_conv.weight'], sd['init_conv.bias'], padding=3)
    r = x
    temb = time_mlp(sd, 'time_mlp.', t, dim)
    hs = []
    for i in range(n_res):
        p = f'downs.{i}.'
        x = resnet_block(sd, p + '0.', x, temb, groups); hs.append(x)
        x = resnet_block(sd, p + '1.', x, temb, groups)
        x = lin_attn(p + '2.', x); hs.append(x)
        if i < n_res - 1:
            b, c, hh, ww = x.shape      # 'b c (h p1) (w p2) -> b (c p1 p2) h w'
            x = x.reshape(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, hh // 2, ww // 2)
            x = F.conv2d(x, sd[p + '3.1.weight'], sd[p + '3.1.bias'])
        else:
            x = F.conv2d(x, sd[p + '3.weight'], sd[p + '3.bias'], padding=1)
    x = resnet_block(sd, 'mid_block1.', x, temb, groups)
    # full attention, unet.py:240-259
    y = channel_layernorm(x, sd['mid_attn.fn.norm.g'])
    b, c, hh, ww = y.shape
    q, k, v = F.conv2d(y, sd['mid_attn.fn.fn.to_qkv.weight']).chunk(3, dim=1)
    q, k, v = (z.reshape(b, heads, dim_head, hh * ww) for z in (q, k, v))
    sim = torch.einsum('bhdi,bhdj->bhij', q * dim_head ** -0.5, k)
    o = torch.einsum('bhij,bhdj->bhid', sim.softmax(dim=-1), v)
    o = o.permute(0, 1, 3, 2).reshape(b, heads * dim_head, hh, ww)
    x = F.conv2d(o, sd['mid_attn.fn.fn.to_out.weight'], sd['mid_attn.fn.fn.to_out.bias']) + x
    x = resnet_block(sd, 'mid_block2.', x, temb, groups)
    for i in range(n_res):
        p = f'ups.{i}.'
        x = resnet_block(sd, p + '0.', torch.cat([x, hs.pop()], dim=1), temb, groups)
        x = resnet_block(sd, p + '1.', torch.cat([x, hs.pop()], dim=1), temb, groups)
        x = lin_attn(p + '2.', x)
        if i < n_res - 1:
            x = F.interpolate(x, scale_factor=2, mode='nearest')
            x = F.conv2d(x, sd[p + '3.1.weight'], sd[p + '3.1.bias'], padding=1)
        else:
            x = F.conv2d(x, sd[p + '3.weight'], sd[p + '3.bias'], padding=1)
    x = resnet_block(sd, 'final_res_block.', torch.cat([x, r], dim=1), temb, groups)
    return F.conv2d(x, sd['final_conv.weight'], sd['final_conv.bias'])


# ----------------------------------------------------------------------------- smoke Unet3D_with_Conv3D
def relative_position_bucket(rel, num_buckets=32, max_distance=32):
    """conv3d.py:86-104 (T5-style bidirectional buckets). rel = k_pos - q_pos (int64 tensor)."""
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return ret + torch.where(n < max_exact, n, large)


def time_rel_pos_bias(emb_w, n):
    """conv3d.py:106-112 -> [heads, n, n]."""
    pos = torch.arange(n)
    rel = pos[None, :] - pos[:, None]
    return emb_w[relative_position_bucket(rel)].permute(2, 0, 1)


def rotary(t, freqs):
    """rotary_embedding_torch (unpinned third party) as used at conv3d.py:320-322: interleaved pairs."""
    n = t.shape[-2]
    ang = (torch.arange(n, dtype=freqs.dtype)[:, None] * freqs[None, :]).repeat_interleave(2, dim=-1)
    x = t.reshape(*t.shape[:-1], -1, 2)
    rot = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(t.shape)
    return t * ang.cos() + rot * ang.sin()


def token_attention(x, w_qkv, w_out, heads, dim_head, freqs=None, pos_bias=None):
    """conv3d.py:294-353 on [..., n, C] (focus_present_mask all-False => no masking)."""
    q, k, v = F.linear(x, w_qkv).chunk(3, dim=-1)
    split = lambda z: z.reshape(*z.shape[:-1], heads, dim_head).transpose(-2, -3)
    q, k, v = split(q), split(k), split(v)
    q = q * dim_head ** -0.5
    if freqs is not None:
        q, k = rotary(q, freqs), rotary(k, freqs)
    sim = q @ k.transpose(-1, -2)
    if pos_bias is not None:
        sim = sim + pos_bias
    sim = sim - sim.amax(dim=-1, keepdim=True).detach()
    o = sim.softmax(dim=-1) @ v
    o = o.transpose(-2, -3).reshape(*x.shape[:-1], heads * dim_head)
    return F.linear(o, w_out)


def unet3d_forward(sd, x, t, *, dim, dim_mults=(1, 2, 4), groups=8, heads=4, dim_head=32):
    """conv3d.py:487-574 with cond=None. x is [B, F, C, H, W] and so is the result."""
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n_res = len(in_out)
    x = x.permute(0, 2, 1, 3, 4)
    bias = time_rel_pos_bias(sd['time_rel_pos_bias.relative_attention_bias.weight'], x.shape[2])

    def temporal(pfx, x):
        y = channel_layernorm(x, sd[pfx + 'fn.norm.gamma'])
        b, c, f, hh, ww = y.shape
        y = y.permute(0, 3, 4, 2, 1).reshape(b, hh * ww, f, c)
        y = token_attention(y, sd[pfx + 'fn.fn.fn.to_qkv.weight'], sd[pfx + 'fn.fn.fn.to_out.weight'], heads, dim_head,
                            freqs=sd[pfx + 'fn.fn.fn.rotary_emb.freqs'], pos_bias=bias)
        return y.reshape(b, hh, ww, f, c).permute(0, 4, 3, 1, 2) + x

    def spatial_linear(pfx, x):
        y = channel_layernorm(x, sd[pfx + 'fn.norm.gamma'])
        b, c, f, hh, ww = y.shape
        y = y.permute(0, 2, 1, 3, 4).reshape(b * f, c, hh, ww)
        y = linear_attention_2d(y, sd[pfx + 'fn.fn.to_qkv.weight'], sd[pfx + 'fn.fn.to_out.weight'], sd[pfx + 'fn.fn.to_out.bias'],
                                heads, dim_head)
        return y.reshape(b, f, c, hh, ww).permute(0, 2, 1, 3, 4) + x

    k0 = sd['init_conv.weight'].shape[-1]
    x = F.conv3d(x, sd['init_conv.weight'], sd['init_conv.bias'], padding=k0 // 2)
    x = temporal('init_temporal_attn.', x)
    r = x
    temb = time_mlp(sd, 'time_mlp.', t, dim)
    hs = []
    for i in range(n_res):
        p = f'downs.{i}.'
        x = resnet_block(sd, p + '0.', x, temb, groups)
        x = resnet_block(sd, p + '1.', x, temb, groups)
        x = spatial_linear(p + '2.', x)
        x = temporal(p + '3.', x)
        hs.append(x)
        if i < n_res - 1:
            x = F.conv3d(x, sd[p + '4.weight'], sd[p + '4.bias'], stride=(1, 2, 2), padding=(0, 1, 1))
    x = resnet_block(sd, 'mid_block1.', x, temb, groups)
    y = channel_layernorm(x, sd['mid_spatial_attn.fn.norm.gamma'])
    b, c, f, hh, ww = y.shape
    y = y.permute(0, 2, 3, 4, 1).reshape(b, f, hh * ww, c)
    y = token_attention(y, sd['mid_spatial_attn.fn.fn.fn.to_qkv.weight'], sd['mid_spatial_attn.fn.fn.fn.to_out.weight'], heads, dim_head)
    x = y.reshape(b, f, hh, ww, c).permute(0, 4, 1, 2, 3) + x
    x = temporal('mid_temporal_attn.', x)
    x = resnet_block(sd, 'mid_block2.', x, temb, groups)
    for i in range(n_res):
        p = f'ups.{i}.'
        x = torch.cat([x, hs.pop()], dim=1)
        x = resnet_block(sd, p + '0.', x, temb, groups)
        x = resnet_block(sd, p + '1.', x, temb, groups)
        x = spatial_linear(p + '2.', x)
        x = temporal(p + '3.', x)
        if i < n_res - 1:
            x = F.conv_transpose3d(x, sd[p + '4.weight'], sd[p + '4.bias'], stride=(1, 2, 2), padding=(0, 1, 1))
    x = torch.cat([x, r], dim=1)
    x = resnet_block(sd, 'final_conv.0.', x, None, groups)
    x = F.conv3d(x, sd['final_conv.1.weight'], sd['final_conv.1.bias'])
    return x.permute(0, 2, 1, 3, 4)
