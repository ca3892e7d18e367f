import sys, torch
sys.path.insert(0, '.')
from wdno_amd import wavelets, _lib
lib = _lib.load()
x3 = torch.randn(32, 32, 64, 64, device='cuda')
c3 = wavelets.dwt_packed(x3, 'bior1.3', 'zero', 3)
fn = lambda: wavelets.idwt_packed(c3, 'bior1.3', 'zero', 3)
for mode in (0, 12, 13, 14):
    lib.wdno_set_debug(mode)
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print('debug', mode, 'us per launch', e0.elapsed_time(e1) / 200 * 1e3)
lib.wdno_set_debug(0)
