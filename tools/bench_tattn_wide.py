"""Times Residual(PreNorm(temporal attention)) and Residual(PreNorm(SpatialLinearAttention)) of the 128- / 256-channel levels: fused
(csrc/attn_fused_wide.hip, csrc/linattn_fused_wide.hip) vs layer by layer, no gradients (a sampling step). Usage: python tools/bench_tattn_wide.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wdno_amd import ops, tree_path  # noqa: E402

for t in ('third_party', 'smoke', 'burgers'):
    sys.path.insert(0, tree_path(t))
from video_diffusion_pytorch import video_diffusion_pytorch_conv3d as V  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    torch.manual_seed(0)
    for c, b, h, w in ((128, 8, 20, 20), (256, 8, 10, 10), (128, 8, 10, 10), (64, 8, 20, 20), (64, 8, 40, 40)):
        rot = V.RotaryEmbedding(32)
        att = V.EinopsToAndFrom('b c f h w', 'b (h w) f c', V.Attention(c, heads=4, dim_head=32, rotary_emb=rot))
        blk = V.Residual(V.PreNorm(c, att)).cuda()
        rpb = V.RelativePositionBias(heads=4, max_distance=32).cuda()
        x = torch.randn(b, 24, h, w, c, device='cuda')
        with torch.no_grad():
            bias = rpb(24, device='cuda')
            t_f = timed(lambda: blk(x, pos_bias=bias))
            ops.FUSED_TATTN_WIDE = False
            ops.FUSED_TATTN = c != 64
            t_l = timed(lambda: blk(x, pos_bias=bias))
            ops.FUSED_TATTN_WIDE = True
            ops.FUSED_TATTN = True
        print(f'[{b},24,{h},{w},{c}]: fused {t_f:7.1f} us   layer by layer {t_l:7.1f} us', flush=True)


def main_linear():
    torch.manual_seed(0)
    for c, b, h, w in ((128, 8, 20, 20), (256, 8, 10, 10), (128, 8, 10, 10), (64, 8, 20, 20), (64, 8, 40, 40)):
        blk = V.Residual(V.PreNorm(c, V.SpatialLinearAttention(c, heads=4))).cuda()
        x = torch.randn(b, 24, h, w, c, device='cuda')
        with torch.no_grad():
            t_f = timed(lambda: blk(x))
            ops.FUSED_LATTN_WIDE = False
            ops.FUSED_LATTN = c != 64
            t_l = timed(lambda: blk(x))
            ops.FUSED_LATTN_WIDE = True
            ops.FUSED_LATTN = True
        print(f'linear attention [{b},24,{h},{w},{c}]: fused {t_f:7.1f} us   layer by layer {t_l:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
    main_linear()
