"""Temporal attention (24 frames, rotary + relative-position bias) at the three U-Net levels: old thread-per-row kernels
(debug 5) vs the one-wave-per-item MFMA kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wdno_amd import ops
lib = ops._lib_()
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
heads, n = 4, 24
for hw in (1600, 400, 100):
    B = 8
    R = B * n * hw
    qkv = torch.randn(R, 3 * heads * 32, device='cuda', requires_grad=True)
    bias = torch.randn(heads, n, n, device='cuda', requires_grad=True)
    freqs = 1.0 / (10000.0 ** (torch.arange(0, 32, 2, device='cuda').float() / 32))
    rot = ops.rotary_tables(freqs, n)
    go = torch.randn(R, heads * 32, device='cuda')
    args = dict(heads=heads, n_uo=B, n_ui=hw, n_tok=n, so=n * hw, si=1, st=hw, scale=32 ** -0.5, bias=bias, rot=rot)
    for mode in (44, 0, 44, 0):
        lib.wdno_set_debug(mode)
        tf = timeit(lambda: ops.softmax_attention(qkv.detach(), **args))
        out = ops.softmax_attention(qkv, **args)
        def bw():
            qkv.grad = None; bias.grad = None
            out.backward(go, retain_graph=True)
        tb = timeit(bw)
        name = {44: 'generic n', 0: 'default  '}[mode]
        print(f'hw={hw:5d} mode={name} fwd {tf:7.1f} us  bwd {tb:7.1f} us')
    lib.wdno_set_debug(0)
