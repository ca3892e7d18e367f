"""Weight-gradient kernels with items dealt round-robin over the blocks (debug 6) vs one contiguous item range per XCD."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wdno_amd import ops
lib = ops._lib_()
CASES = [
    ('l0 3x3x3 64->64', (8, 24, 40, 40, 64), (64, 64, 3, 3, 3)),
    ('l0 3x3x3 128->64', (8, 24, 40, 40, 128), (64, 128, 3, 3, 3)),
    ('l1 3x3x3 128->128', (8, 24, 20, 20, 128), (128, 128, 3, 3, 3)),
    ('l1 3x3x3 256->64', (8, 24, 20, 20, 256), (64, 256, 3, 3, 3)),
    ('l2 3x3x3 256->256', (8, 24, 10, 10, 256), (256, 256, 3, 3, 3)),
]
iters = int(os.environ.get('ITERS', '10'))
def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, xs, ws in CASES:
    x = torch.randn(*xs, device='cuda')
    k, c = ws[0], ws[1]
    y = torch.randn(*xs[:4], k, device='cuda')
    xpl = ops.split_f16(x.reshape(-1, xs[-1]))
    ypl = ops.split_f16(y.reshape(-1, k))
    flops = 2.0 * y.numel() * c * 27
    out = []
    for mode in (int(os.environ.get('MODE_A', '6')), 0, int(os.environ.get('MODE_A', '6')), 0):
        lib.wdno_set_debug(mode)
        t = timeit(lambda: ops.conv_wgrad_h3(xpl, tuple(xs[:4]), ypl, tuple(xs[1:4]), (3, 3, 3), (1, 1, 1), (1, 1, 1)))
        out.append(f'm{mode} {t:6.3f} ms {flops / t / 1e9:6.1f} TF/s')
    lib.wdno_set_debug(0)
    print(f'{name:20s} ' + ' | '.join(out), flush=True)
