"""Micro-benchmark of the implicit-GEMM convolution kernels on the geometries of the smoke U-Net (batch 8).
Columns: exact-fp32 MFMA kernels and the 3 x fp16-split kernels (split pre-pass timed separately)."""
import os, sys
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
    ('burgers l0 3x3 128->128', (16, 1, 64, 64, 128), (128, 128, 1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('burgers l1 3x3 256->256', (16, 1, 32, 32, 256), (256, 256, 1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('burgers l2 3x3 512->512', (16, 1, 16, 16, 512), (512, 512, 1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('burgers l3 3x3 1024->1024', (16, 1, 8, 8, 1024), (1024, 1024, 1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('burgers d2 3x3 256->256 @16', (16, 1, 16, 16, 256), (256, 256, 1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('burgers d3 3x3 512->512 @8', (16, 1, 8, 8, 512), (512, 512, 1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ('burgers u3 3x3 1536->1024 @8', (16, 1, 8, 8, 1536), (1024, 1536, 1, 3, 3), (1, 1, 1), (0, 1, 1)),
]
only = [a for a in sys.argv[1:] if not a.startswith('-')]
fp32_too = '--fp32' in sys.argv


def timeit(fn, iters=10):
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
    osp = tuple(y.shape[1:4])
    P = y.numel() // kp
    flops = 2.0 * P * k * c * ks[0] * ks[1] * ks[2]
    tf = lambda t: flops / t / 1e9
    line = f'{name:22s}'
    if fp32_too:
        t_f = timeit(lambda: ops.conv_fwd_raw(x, wp, None, None, ks, st, pd, kp))
        t_w = timeit(lambda: ops.conv_wgrad_raw(x, y, ks, st, pd))
        line += f' f32: fwd {tf(t_f):6.1f} wgrad {tf(t_w):6.1f} |'
    t_s = timeit(lambda: ops.split_f16(x.reshape(-1, cp)))
    xpl = ops.split_f16(x.reshape(-1, cp))
    ypl = ops.split_f16(y.reshape(-1, kp))
    t_hf = timeit(lambda: ops.conv_fwd_h3(xpl, tuple(xs[:4]), w, ops.pack_fwd, 'f', None, None, ks, st, pd, kp))
    if '--cold' in sys.argv:          # rotate over input copies larger than the 256 MB memory-side cache: every launch reads x from HBM
        nrot = max(2, int(600e6 // (xpl[0].numel() * 4)) + 1)
        rot = [(xpl[0].clone(), xpl[1].clone(), xpl[2]) for _ in range(nrot)]
        outs = [torch.empty_like(y) for _ in range(nrot)]
        it = [0]
        def cold():
            i = it[0] = (it[0] + 1) % nrot
            ops.conv_fwd_h3(rot[i], tuple(xs[:4]), w, ops.pack_fwd, 'f', None, None, ks, st, pd, kp, out=outs[i])
        t_c = timeit(cold, iters=2 * nrot)
        print(f'   cold inputs ({nrot} rotating copies): fwd {t_c:.3f} ms vs {t_hf:.3f} ms warm')
        del rot, outs
    if '--ablate' in sys.argv:
        lib = ops._lib_()
        f = lambda: ops.conv_fwd_h3(xpl, tuple(xs[:4]), w, ops.pack_fwd, 'f', None, None, ks, st, pd, kp)
        lib.wdno_set_debug(21); t1 = timeit(f)
        lib.wdno_set_debug(0)
        print(f'   ablation fwd: full {t_hf:.3f} ms | compute waves alone (no DMA issue) {t1:.3f} ms')
    if '--stamps' in sys.argv:
        lib = ops._lib_()
        for base, what in ((0, 'fwd'), (100, 'fwd, DMA issue off')):
            vals = {}
            for mode in (23, 24, 26):
                lib.wdno_set_debug(base + mode)
                rec = ops._amax_slot(x.device)
                for _ in range(5):
                    rec.zero_()
                    ops.conv_fwd_h3(xpl, tuple(xs[:4]), w, ops.pack_fwd, 'f', None, None, ks, st, pd, kp, amax_rec=rec)
                torch.cuda.synchronize()
                vals[mode] = float(rec.max())
            lib.wdno_set_debug(0)
            print(f'   stamps {what} (slowest compute wave): {vals[23]:.0f} shader cycles in the kernel, {vals[24]:.0f} of them in tile epilogues, '
                  f'{vals[26] * 10:.0f} ns wall -> {vals[23] / (vals[26] * 10):.2f} GHz')
    t_hw = timeit(lambda: ops.conv_wgrad_h3(xpl, tuple(xs[:4]), ypl, osp, ks, st, pd))
    if '--ablate' in sys.argv:
        lib = ops._lib_()
        fw = lambda: ops.conv_wgrad_h3(xpl, tuple(xs[:4]), ypl, osp, ks, st, pd)
        lib.wdno_set_debug(21); t1 = timeit(fw)
        lib.wdno_set_debug(22); t2 = timeit(fw)
        lib.wdno_set_debug(0)
        print(f'   ablation wgrad: full {t_hw:.3f} ms | all pieces out of range (issued, nothing travels) {t1:.3f} ms | no piece instructions {t2:.3f} ms')
    line += f' h3: fwd {t_hf:6.3f} ms {tf(t_hf):6.1f} TF/s | wgrad {t_hw:6.3f} ms {tf(t_hw):6.1f} TF/s'
    if st == (1, 1, 1):
        pdd = tuple(kk - 1 - p for kk, p in zip(ks, pd))
        t_hd = timeit(lambda: ops.conv_fwd_h3(ypl, tuple(y.shape[:4]), w, lambda w_, c8_, k_: ops.pack_dgrad(w_, k_, c8_), 'd', None, None, ks, (1, 1, 1), pdd, cp))
        line += f' | dgrad {t_hd:6.3f} ms {tf(t_hd):6.1f} TF/s'
    line += f' | split(x) {t_s:6.3f} ms'
    print(line, flush=True)
