"""Level-0 temporal attention block of the smoke U-Net at the bench size ([8, 24, 40, 40, 64]): one fused launch (csrc/attn_fused.hip) vs the
layer-by-layer forward (LayerNorm planes -> to_qkv -> attention -> to_out + residual), HIP-event timed under no_grad.

    python tools/bench_tattn.py [batch] [hw_side]"""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wdno_amd import ops, tree_path  # noqa: E402
for t in ('third_party', 'smoke', 'burgers'):
    sys.path.insert(0, tree_path(t))
from video_diffusion_pytorch import video_diffusion_pytorch_conv3d as V  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
side = int(sys.argv[2]) if len(sys.argv) > 2 else 40
torch.manual_seed(0)
rot = V.RotaryEmbedding(32)
blk = V.Residual(V.PreNorm(64, V.EinopsToAndFrom('b c f h w', 'b (h w) f c', V.Attention(64, heads=4, dim_head=32, rotary_emb=rot)))).cuda()
rpb = V.RelativePositionBias(heads=4, max_distance=32).cuda()
x = torch.randn(b, 24, side, side, 64, device='cuda')


def timed(n=20):
    for _ in range(3):
        y = blk(x, pos_bias=bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = blk(x, pos_bias=bias)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, y


with torch.no_grad():
    bias = rpb(24, device='cuda')
    t_f, y_f = timed()
    ops.FUSED_TATTN = False
    t_l, y_l = timed()
    ops.FUSED_TATTN = True
mb = 2 * x.numel() * 4 / 1e6
print(f'[{b},24,{side},{side},64]: fused {t_f:.1f} us ({mb / t_f * 1e-3 * 1e3:.2f} GB/s-equivalent: {mb:.0f} MB in+out -> {mb / t_f:.3f} TB/s), '
      f'layer by layer {t_l:.1f} us, max |diff| {(y_f - y_l).abs().max().item():.2e}')


# forward + backward (a training step's use of the block): one launch each way vs the layer-by-layer path
def timed_train(n=10):
    xr = x.clone().requires_grad_(True)
    gy = torch.randn(x.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(1))
    fw = bw = 0.0
    for it in range(n + 3):
        for p_ in list(blk.parameters()) + list(rpb.parameters()):
            p_.grad = None
        xr.grad = None
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        y = blk(xr, pos_bias=rpb(24, device='cuda'))
        e1.record()
        y.backward(gy)
        e2.record()
        torch.cuda.synchronize()
        if it >= 3:
            fw += e0.elapsed_time(e1) * 1e3 / n
            bw += e1.elapsed_time(e2) * 1e3 / n
    return fw, bw, xr.grad.clone(), blk.fn.fn.fn.to_qkv.weight.grad.clone()


f_f, b_f, dx_f, dw_f = timed_train()
ops.FUSED_TATTN_BWD = False
f_l, b_l, dx_l, dw_l = timed_train()
ops.FUSED_TATTN_BWD = True
print(f'with gradients: fused forward {f_f:.1f} us + backward {b_f:.1f} us (3 x {x.numel() * 4 / 1e6:.0f} MB in+out -> {3 * x.numel() * 4 / 1e6 / b_f:.3f} TB/s); '
      f'layer by layer forward {f_l:.1f} us + backward {b_l:.1f} us; max |d dx| {(dx_f - dx_l).abs().max().item():.2e} of {dx_l.abs().max().item():.2e}, '
      f'max |d dW_qkv| {(dw_f - dw_l).abs().max().item():.2e} of {dw_l.abs().max().item():.2e}')
