import sys, torch
sys.path.insert(0, '.')
from wdno_amd import ops
def t(fn, it=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for rows, c, k in ((16, 512, 2048), (16, 512, 256), (16, 128, 512), (8, 256, 128), (8, 256, 512)):
    x = torch.randn(rows, c, device='cuda', requires_grad=True)
    w = torch.randn(k, c, device='cuda', requires_grad=True)
    b = torch.randn(k, device='cuda', requires_grad=True)
    f = t(lambda: ops.conv_cl(x.detach(), w.detach(), b.detach()))
    y = ops.conv_cl(x, w, b)
    g = torch.randn_like(y)
    def bw():
        x.grad = w.grad = b.grad = None
        y.backward(g, retain_graph=True)
    print(rows, c, k, f'fwd {f:.1f} us  bwd(dgrad+wgrad) {t(bw):.1f} us')
