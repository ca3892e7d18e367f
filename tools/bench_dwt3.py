"""3-D synthesis kernels A/B: 20 launches captured in one graph (no host gaps), per debug mode."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wdno_amd import wavelets, ops
lib = ops._lib_()
x3 = torch.randn(32, 32, 64, 64, device='cuda')
c3 = wavelets.dwt_packed(x3, 'bior1.3', 'zero', 3)
xs = torch.randn(10, 64, 128, 128, device='cuda')
cs = wavelets.dwt_packed(xs, 'bior1.3', 'zero', 3)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 / n * 1e3
for rep in range(2):
    for mode, name in ((45, 'all frames in LDS (r2)'), (0, 'streaming (r3)')):
        lib.wdno_set_debug(mode)
        a = t(lambda: wavelets.idwt_packed(c3, 'bior1.3', 'zero', 3))
        b = t(lambda: wavelets.idwt_packed(cs, 'bior1.3', 'zero', 3))
        print(f'{name:26s} [32,8,18,34,34] -> [32,32,64,64]: {a:6.2f} us   [10,8,34,66,66] -> [10,64,128,128]: {b:6.2f} us')
lib.wdno_set_debug(0)
r0 = wavelets.idwt_packed(c3, 'bior1.3', 'zero', 3)
lib.wdno_set_debug(45)
r1 = wavelets.idwt_packed(c3, 'bior1.3', 'zero', 3)
lib.wdno_set_debug(11)
r2 = wavelets.idwt_packed(c3, 'bior1.3', 'zero', 3)
lib.wdno_set_debug(0)
print('streaming == r2 fused:', torch.equal(r0, r1), ' == per-axis:', torch.equal(r0, r2), ' max |x - idwt(dwt(x))|:', float((r0[:, :32, :64, :64] - x3).abs().max()))
