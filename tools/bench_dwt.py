"""DWT / IDWT kernel time on the two BASELINE.json shapes (HIP events around graph replays of 20 launches, inputs resident in HBM).
Algorithmic bytes = input + output of one transform: 2-D per/bior2.4 [64,2,160,128]: 20.97 MB; 3-D zero/bior1.3
[32,32,64,64]: 38.1 MB."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wdno_amd import wavelets  # noqa: E402


def timed(fn, n=20):
    """us per launch: n launches captured in one graph (no host gaps between them), 10 replays."""
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 / n * 1e3


def run():
    out = {}
    x2 = torch.randn(64, 2, 160, 128, device='cuda')
    c2 = wavelets.dwt_packed(x2, 'bior2.4', 'periodization', 2)
    x3 = torch.randn(32, 32, 64, 64, device='cuda')
    c3 = wavelets.dwt_packed(x3, 'bior1.3', 'zero', 3)
    cases = {
        'dwt2_per_bior2.4_fwd': (lambda: wavelets.dwt_packed(x2, 'bior2.4', 'periodization', 2), x2.numel() * 4 + c2.numel() * 4),
        'dwt2_per_bior2.4_inv': (lambda: wavelets.idwt_packed(c2, 'bior2.4', 'periodization', 2), x2.numel() * 4 + c2.numel() * 4),
        'dwt3_zero_bior1.3_fwd': (lambda: wavelets.dwt_packed(x3, 'bior1.3', 'zero', 3), x3.numel() * 4 + c3.numel() * 4),
        'dwt3_zero_bior1.3_inv': (lambda: wavelets.idwt_packed(c3, 'bior1.3', 'zero', 3), x3.numel() * 4 + c3.numel() * 4),
    }
    for k, (fn, nbytes) in cases.items():
        us = timed(fn)
        out[k] = {'us': round(us, 2), 'MB': round(nbytes / 1e6, 2), 'GB/s': round(nbytes / us / 1e3, 1), 'frac_of_8TBs': round(nbytes / us / 1e3 / 8000, 4)}
    return out


if __name__ == '__main__':
    print(json.dumps(run(), indent=1))
