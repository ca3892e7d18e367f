"""What would Winograd F(2,3) along W cost on this hardware? An emulation with the EXISTING kernels, no Winograd arithmetic:
F(2,3) along W turns a 3 x 3 x 3 convolution over W output columns into FOUR independent 3 x 3 x 1 convolutions (one per transformed
position xi, each with its own weights) over W / 2 column pairs: 4 x (9 C) x (P / 2) multiply-adds instead of (27 C) x P = 2/3 of the
matrix work, but 2x the operand bytes (4 transformed values per 2 inputs) and no reuse of a staged tile across the three W taps.
Timed here: one 3x3x3 launch on [8,24,40,40,C] against four 3x3x1 launches on [8,24,40,20,C] (same fp32-equivalent split kernels, random
operands: the power-limited regime of DESIGN.md section 6), plus the two extra passes a real implementation needs -- an input transform
that writes 2x the planes and an output transform -- approximated by split passes over tensors of those sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wdno_amd import ops


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for c, k, sp in ((64, 64, (8, 24, 40, 40)), (128, 128, (8, 24, 20, 20)), (256, 256, (8, 24, 10, 10))):
    n, d, h, w = sp
    x = torch.randn(n, d, h, w, c, device='cuda')
    w3 = torch.randn(k, c, 3, 3, 3, device='cuda') * 0.05
    xpl = ops.split_f16(x.reshape(-1, c))
    t_full = timeit(lambda: ops.conv_fwd_h3(xpl, (n, d, h, w), w3, ops.pack_fwd, 'f', None, None, (3, 3, 3), (1, 1, 1), (1, 1, 1), k))
    xh = torch.randn(n, d, h, w // 2, c, device='cuda')
    xhp = ops.split_f16(xh.reshape(-1, c))
    ws = [torch.randn(k, c, 3, 3, 1, device='cuda') * 0.05 for _ in range(4)]

    def four():
        for wq in ws:
            ops.conv_fwd_h3(xhp, (n, d, h, w // 2), wq, ops.pack_fwd, 'f', None, None, (3, 3, 1), (1, 1, 1), (1, 1, 0), k)
    t_four = timeit(four)
    # input transform: reads x (fp32), writes 4 transformed values per input pair as (hi, lo) planes = 2x the planes of x
    x2 = torch.randn(n * d * h * w * 2, c, device='cuda')
    t_in = timeit(lambda: ops.split_f16(x2)) * 0.75              # (reads 1x, writes 2x: 3 units against the 4 of a split of a 2x tensor)
    # output transform: reads 4 fp32 partial outputs per pair (2x y), writes y -- unless the four products share one kernel (then free)
    t_out = timeit(lambda: ops.split_f16(x2)) * 0.75
    print(f'C={c:3d} K={k:3d} [{n},{d},{h},{w}]: 3x3x3 {t_full * 1e3:7.1f} us | four 3x3x1 on W/2 {t_four * 1e3:7.1f} us ({t_four / t_full:.2f} of it; matrix work 0.67) '
          f'| + input transform ~{t_in * 1e3:5.1f} us, output transform (if not fused) ~{t_out * 1e3:5.1f} us '
          f'-> {(t_four + t_in) / t_full:.2f} .. {(t_four + t_in + t_out) / t_full:.2f} of the direct kernel')
