"""Spatial linear attention (4 heads x 32, tokens = pixels of one frame) at the three U-Net levels: forward and backward time,
and the bytes each must move at least (qkv read + out written; backward: qkv, dout read + dqkv written)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wdno_amd import ops
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
heads = 4
for hw in (1600, 400, 100):
    units = 8 * 24
    R = units * hw
    qkv = torch.randn(R, 3 * heads * 32, device='cuda', requires_grad=True)
    go = torch.randn(R, heads * 32, device='cuda')
    tf = timeit(lambda: ops.linear_attention(qkv.detach(), units, hw, heads, 32 ** -0.5))
    out = ops.linear_attention(qkv, units, hw, heads, 32 ** -0.5)
    def bw():
        qkv.grad = None
        out.backward(go, retain_graph=True)
    tb = timeit(bw)
    fb, bb = R * (384 + 128) * 4, R * (384 + 128 + 384) * 4
    print(f'hw={hw:5d} fwd {tf:7.1f} us ({fb / tf / 1e6:5.2f} TB/s of the minimum traffic)  bwd {tb:7.1f} us ({bb / tb / 1e6:5.2f} TB/s)')
