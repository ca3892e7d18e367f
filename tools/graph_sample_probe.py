"""Diagnostic: capture one DDPM p_sample step of the smoke model in a HIP graph and report where capture breaks."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import _lib, ops, diffusion_core as K
_lib.load()
dev = torch.device('cuda', 0)
B = int(os.environ.get('B', '8'))
dif = bench.build_model(dev, B)
x = torch.randn(B, 24, 42, 40, 40, device=dev)
init = torch.randn(B, 24, 40, 40, device=dev)
control = torch.randn(B, 24, 16, 40, 40, device=dev)
desc = dif._desc(tuple(x.shape), dif.padded_shape)
src = dif._condition_source(tuple(x.shape), dev, init, control, None)
bt = torch.full((B,), 500, device=dev, dtype=torch.long)
noise = torch.randn_like(x)

def step(xx):
    eps = dif.model(xx, bt, None)
    xn, xs = K.p_sample_update(dif, xx, eps, noise, bt, clamp=True)
    return K.apply_cond(xn, src, desc)

with torch.no_grad():
    for _ in range(3):
        y = step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        y = step(x)
    torch.cuda.synchronize()
    print('eager ms/step', (time.perf_counter() - t0) / 10 * 1e3, flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            y = step(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            y = step(x)
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        print('graph ms/step', (time.perf_counter() - t0) / 10 * 1e3, flush=True)
        ye = step(x)
        print('graph == eager:', torch.equal(y, ye), float((y - ye).abs().max()))
    except Exception:
        traceback.print_exc()
