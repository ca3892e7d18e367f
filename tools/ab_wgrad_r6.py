"""Round-6 A/B of the window weight-gradient kernel (csrc/conv_wgrad_h3d.hip: conv_wgrad_h3w_kernel): split-pair mode (the odd tap row paired
across two pixel splits instead of with an empty window) against the round-5 item plan (library debug mode 70), same process, alternating
repetitions, on the smoke / Burgers layer geometries; and the two plans' results against each other (fp32 sums in another order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wdno_amd import ops

CASES = [
    ('l0 3x3x3 64->64', (8, 24, 40, 40, 64), (64, 64, 3, 3, 3)),
    ('l0 3x3x3 128->64', (8, 24, 40, 40, 128), (64, 128, 3, 3, 3)),
    ('l1 3x3x3 64->128', (8, 24, 20, 20, 64), (128, 64, 3, 3, 3)),
    ('l1 3x3x3 128->128', (8, 24, 20, 20, 128), (128, 128, 3, 3, 3)),
    ('l1 3x3x3 256->128', (8, 24, 20, 20, 256), (128, 256, 3, 3, 3)),
    ('l2 3x3x3 256->256', (8, 24, 10, 10, 256), (256, 256, 3, 3, 3)),
    ('l2 3x3x3 512->256', (8, 24, 10, 10, 512), (256, 512, 3, 3, 3)),
    ('burgers l0 3x3 128->128', (16, 1, 64, 64, 128), (128, 128, 1, 3, 3)),
    ('burgers l1 3x3 256->256', (16, 1, 32, 32, 256), (256, 256, 1, 3, 3)),
    ('burgers l0 3x3 128->128 b256', (256, 1, 64, 64, 128), (128, 128, 1, 3, 3)),
]
lib = ops._lib_()
only = [a for a in sys.argv[1:] if not a.startswith('-')]
if only:
    CASES = [c for c in CASES if any(o in c[0] for o in only)]
bf16 = '--bf16' in sys.argv
if bf16:
    ops.CONV_MATH = 'bf16'


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, xs, ws in CASES:
    x = torch.randn(*xs, device='cuda')
    k, c = ws[0], ws[1]
    ks = tuple(ws[2:])
    pd = tuple(v // 2 for v in ks)
    dy = torch.randn(*xs[:4], k, device='cuda')
    xpl, ypl = ops.split_f16(x.reshape(-1, c)), ops.split_f16(dy.reshape(-1, k))
    f = lambda: ops.conv_wgrad_h3(xpl, tuple(xs[:4]), ypl, tuple(xs[1:4]), ks, (1, 1, 1), pd, param_kc=(k, c))
    res, t = {}, {0: [], 70: []}
    for rep in range(3):
        for mode in (0, 70):
            lib.wdno_set_debug(mode)
            t[mode].append(timeit(f))
            if rep == 0:
                res[mode] = f().clone()
                again = f()
                assert torch.equal(res[mode], again), 'not reproducible'
    lib.wdno_set_debug(0)
    flops = 2.0 * x.numel() // c * k * c * ks[0] * ks[1] * ks[2]
    rel = ((res[0] - res[70]).norm() / res[70].norm()).item()
    a, b = min(t[0]), min(t[70])
    print(f'{name:32s} split-pair {a * 1e3:7.1f} us ({flops / a / 1e9:6.1f} TF/s) | round-5 plan {b * 1e3:7.1f} us ({flops / b / 1e9:6.1f} TF/s) | {a / b:.3f} | rel diff {rel:.2e}'
          f' | all {[round(v * 1e3, 1) for v in t[0]]} vs {[round(v * 1e3, 1) for v in t[70]]}', flush=True)
