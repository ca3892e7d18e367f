"""Same-process A/B of the graph-replayed DDPM sampling step of the smoke base model at batch 8 (and SR DDIM at [2,48,82,80,80] with --sr):
alternating repetitions of library debug modes / ops knobs. Usage: python tools/ab_sampling.py [steps]
  A: default                          B: the attention blocks of the 128- / 256-channel levels layer by layer (ops.FUSED_*_WIDE = False)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wdno_amd import ops, diffusion_core as K  # noqa: E402


def rate(dif, shape, steps, dev):
    x = torch.randn(shape, device=dev)
    b = shape[0]
    init = torch.randn(b, 24, 40, 40, device=dev)
    control = torch.randn(b, 24, 16, 40, 40, device=dev)
    desc = dif._desc(shape, dif.padded_shape)
    src = dif._condition_source(shape, dev, init, control, None)
    with torch.no_grad():
        for _ in range(2):
            x, _ = dif.p_sample(shape, x, 500)
            x = K.apply_cond(x, src, desc)
        sg = K._step_graph(dif, shape, desc, False, False, dev)
        sg.src.copy_(src)
        sg.x.copy_(x)
        for _ in range(3):
            sg.graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            sg.t.fill_(500 - i)
            sg.noise.normal_()
            sg.graph.replay()
        torch.cuda.synchronize()
        r = steps / (time.perf_counter() - t0)
    K._graph_cache.pop(dif, None)
    return r


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = 'cuda'
    dif = bench.build_model(dev)
    shape = (8, 24, 42, 40, 40)
    for rep in range(3):
        for name, wide in (('default', True), ('wide blocks layer by layer', False)):
            ops.FUSED_TATTN_WIDE = ops.FUSED_LATTN_WIDE = wide
            r = rate(dif, shape, steps, dev)
            print(f'rep {rep} {name:28s} {r:7.2f} sample steps/s  {1e3 / r:6.3f} ms', flush=True)
    ops.FUSED_TATTN_WIDE = ops.FUSED_LATTN_WIDE = True


if __name__ == '__main__':
    main()
