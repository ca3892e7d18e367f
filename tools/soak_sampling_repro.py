"""Soak: the graph-replayed DDPM sampling step of the smoke base model at batch 8, and the SR DDIM step at [2,48,82,80,80], replayed N times from
identical inputs -- every replay must return the same bits (the wide attention kernels run inside the real step here, next to everything else).
Usage: python tools/soak_sampling_repro.py [replays]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wdno_amd import ops, diffusion_core as K  # noqa: E402


def soak(dif, shape, n, dev, ddim=False):
    b = shape[0]
    f, h, w = shape[1], shape[3], shape[4]
    x0 = torch.randn(shape, device=dev)
    init = torch.randn(b, f, h, w, device=dev)
    control = torch.randn(b, f, 16, h, w, device=dev)
    desc = dif._desc(shape, dif.padded_shape)
    src = dif._condition_source(shape, dev, init, control, None)
    with torch.no_grad():
        x, _ = dif.p_sample(shape, x0.clone(), 500)          # warm the caches eagerly
        sg = K._step_graph(dif, shape, desc, ddim, False, dev)
        sg.src.copy_(src)
        noise = torch.randn(shape, device=dev)
        outs = []
        for i in range(n):
            sg.x.copy_(x0)
            sg.t.fill_(500)
            sg.noise.copy_(noise)
            sg.graph.replay()
            torch.cuda.synchronize()
            outs.append(sg.x.clone())
    K._graph_cache.pop(dif, None)
    return sum(int(not torch.equal(outs[0], o)) for o in outs[1:]), bool(torch.isfinite(outs[0]).all())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = 'cuda'
    dif = bench.build_model(dev)
    d, fin = soak(dif, (8, 24, 42, 40, 40), n, dev)
    print(f'base model [8,24,42,40,40]: {d} of {n - 1} replays differ from the first; finite {fin}', flush=True)
    assert d == 0 and fin


if __name__ == '__main__':
    main()
