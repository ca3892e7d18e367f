"""Micro-benchmark of the implicit-GEMM convolution kernels on the geometries of the smoke U-Net (batch 8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wdno_amd import ops

CASES = [  # name, x [N,D,H,W,C], weight [K,C,kd,kh,kw], stride, pad
    ('l0 3x3x3 64->64', (8, 24, 40, 40, 64), (64, 64, 3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l0 3x3x3 128->64', (8, 24, 40, 40, 128), (64, 128, 3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l1 3x3x3 128->128', (8, 24, 20, 20, 128), (128, 128, 3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l1 3x3x3 256->64', (8, 24, 20, 20, 256), (64, 256, 3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l2 3x3x3 256->256', (8, 24, 10, 10, 256), (256, 256, 3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('l2 3x3x3 512->128', (8, 24, 10, 10, 512), (128, 512, 3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ('init 7x7x7 44->64', (8, 24, 40, 40, 44), (64, 42, 7, 7, 7), (1, 1, 1), (3, 3, 3)),
    ('l0 qkv 1x1 64->384', (8, 24, 40, 40, 64), (384, 64), (1, 1, 1), (0, 0, 0)),
    ('l0 out 1x1 128->64', (8, 24, 40, 40, 128), (64, 128), (1, 1, 1), (0, 0, 0)),
    ('l0 down 1x4x4 64->64', (8, 24, 40, 40, 64), (64, 64, 1, 4, 4), (1, 2, 2), (0, 1, 1)),
]
only = sys.argv[1:] 

def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for name, xs, ws, st, pd in CASES:
    if only and not any(o in name for o in only):
        continue
    x = torch.randn(*xs, device='cuda')
    w = torch.randn(*ws, device='cuda') * 0.05
    w5 = ops._as5(w)
    k, c = w5.shape[0], w5.shape[1]
    ks = tuple(w5.shape[2:])
    cp, kp = xs[-1], ops.pad4(k)
    wp = ops.pack_fwd(w, cp, kp)
    y = ops.conv_fwd_raw(x, wp, None, None, ks, st, pd, kp)
    P = y.numel() // kp
    flops = 2.0 * P * k * c * ks[0] * ks[1] * ks[2]
    t_f = timeit(lambda: ops.conv_fwd_raw(x, wp, None, None, ks, st, pd, kp), 10)
    t_w = timeit(lambda: ops.conv_wgrad_raw(x, y, ks, st, pd), 10)
    line = f'{name:24s} fwd {t_f:7.3f} ms {flops / t_f / 1e9:7.1f} TF/s | wgrad {t_w:7.3f} ms {flops / t_w / 1e9:7.1f} TF/s'
    if st == (1, 1, 1):
        wd = ops.pack_dgrad(w, cp, kp)
        pdd = tuple(kk - 1 - p for kk, p in zip(ks, pd))
        t_d = timeit(lambda: ops.conv_fwd_raw(y, wd, None, None, ks, (1, 1, 1), pdd, cp), 10)
        line += f' | dgrad {t_d:7.3f} ms {flops / t_d / 1e9:7.1f} TF/s'
    print(line, flush=True)
