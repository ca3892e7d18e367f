import os, sys
sys.path.insert(0, '/root/repo')
import torch
from wdno_amd import ops
lib = ops._lib_()
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
heads, n, hw, B = 4, 24, 1600, 8
R = B * n * hw
qkv = torch.randn(R, 3 * heads * 32, device='cuda')
bias = torch.randn(heads, n, n, device='cuda')
freqs = 1.0 / (10000.0 ** (torch.arange(0, 32, 2, device='cuda').float() / 32))
rot = ops.rotary_tables(freqs, n)
args = dict(heads=heads, n_uo=B, n_ui=hw, n_tok=n, so=n * hw, si=1, st=hw, scale=32 ** -0.5, bias=bias, rot=rot)
for mode, what in ((0, 'full'), (44, 'no softmax'), (45, 'no rotation'), (46, 'no MFMA'), (47, 'no stores'), (48, 'no global loads')):
    lib.wdno_set_debug(mode)
    print(f'{what:16s} {timeit(lambda: ops.softmax_attention(qkv, **args)):7.1f} us')
lib.wdno_set_debug(0)
x = torch.randn(R, 384, device='cuda'); y = torch.empty(R, 128, device='cuda')
print('copy 472 MB read + 157 MB write (torch): ', timeit(lambda: y.copy_(x[:, :128])), 'us (strided)')
z = torch.empty_like(x)
print('contiguous copy 472+472 MB:', timeit(lambda: z.copy_(x)), 'us')
