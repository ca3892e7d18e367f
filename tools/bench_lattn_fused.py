"""The 64-channel SpatialLinearAttention block of the smoke U-Net at the bench size ([8, 24, 40, 40, 64]) under no_grad: two fused passes + a merge
(csrc/linattn_fused.hip) vs the layer-by-layer forward (LayerNorm planes -> to_qkv -> k statistics -> context -> output -> to_out + residual),
HIP-event timed, then the whole sampling step both ways.     python tools/bench_lattn_fused.py [batch] [hw_side]"""
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
blk = V.Residual(V.PreNorm(64, V.SpatialLinearAttention(64, heads=4))).cuda()
x = torch.randn(b, 24, side, side, 64, device='cuda')


def timed(n=20):
    for _ in range(3):
        y = blk(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        y = blk(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, y


with torch.no_grad():
    t_f, y_f = timed()
    ops.FUSED_LATTN = False
    t_l, y_l = timed()
    ops.FUSED_LATTN = True
mb = 3 * x.numel() * 4 / 1e6
print(f'[{b},24,{side},{side},64]: fused {t_f:.1f} us ({mb:.0f} MB read twice + written once -> {mb / t_f:.3f} TB/s), layer by layer {t_l:.1f} us, '
      f'max |diff| {(y_f - y_l).abs().max().item():.2e}')



# forward + backward (a training step's use of the block)
def timed_train(n=10):
    xr = x.clone().requires_grad_(True)
    gy = torch.randn(x.shape, device='cuda', generator=torch.Generator('cuda').manual_seed(1))
    fw = bw = 0.0
    for it in range(n + 3):
        for p_ in blk.parameters():
            p_.grad = None
        xr.grad = None
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        y = blk(xr)
        e1.record()
        y.backward(gy)
        e2.record()
        torch.cuda.synchronize()
        if it >= 3:
            fw += e0.elapsed_time(e1) * 1e3 / n
            bw += e1.elapsed_time(e2) * 1e3 / n
    return fw, bw, xr.grad.clone(), blk.fn.fn.to_qkv.weight.grad.clone()


f_f, b_f, dx_f, dw_f = timed_train()
ops.FUSED_LATTN_BWD = False
f_l, b_l, dx_l, dw_l = timed_train()
ops.FUSED_LATTN_BWD = True
print(f'with gradients: fused forward {f_f:.1f} us + backward {b_f:.1f} us; layer by layer forward {f_l:.1f} us + backward {b_l:.1f} us; '
      f'max |d dx| {(dx_f - dx_l).abs().max().item():.2e} of {dx_l.abs().max().item():.2e}, max |d dW_qkv| {(dw_f - dw_l).abs().max().item():.2e} of {dw_l.abs().max().item():.2e}')
if '--block-only' in sys.argv:
    sys.exit(0)

import bench  # noqa: E402
from wdno_amd import diffusion_core as K  # noqa: E402
dif = bench.build_model('cuda', b)
for knob in (True, False, True, False):
    ops.FUSED_LATTN = knob
    r = bench.sampling_leg(dif, 'cuda', b, 30)
    print('FUSED_LATTN', knob, {k: v for k, v in r.items() if k.startswith('batch')})
ops.FUSED_LATTN = True
