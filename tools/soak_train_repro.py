"""Soak: two runs of N training steps (launch by launch) of the smoke base model at the bench size [8,24,42,40,40] from the same seed must leave the
same bits in every parameter (the deterministic reductions + the kernels of round 5 inside the real step). Usage: python tools/soak_train_repro.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wdno_amd import ops  # noqa: E402
from wdno_amd.trainer import TrainStep  # noqa: E402


def run(n, dev):
    torch.manual_seed(0)
    dif = bench.build_model(dev, 8)
    ts = TrainStep(dif, lr=1e-3, use_ema=False)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((8, 24, 42, 40, 40), device=dev, generator=g) * 0.5
    torch.manual_seed(7)              # the step's own draws (t, noise)
    losses = []
    for _ in range(n):
        losses.append(float(ts.step(x)[0]))
    torch.cuda.synchronize()
    flat = ts.opt.buf.flat_param.detach().clone()
    del ts, dif
    ops.drop_weight_caches()
    torch.cuda.empty_cache()
    return flat, losses


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device('cuda', 0)
    a, la = run(n, dev)
    b, lb = run(n, dev)
    nd = int((a != b).sum())
    print(f'{n} steps twice: {nd} of {a.numel()} parameters differ; losses equal: {la == lb}; last loss {la[-1]:.6f}', flush=True)
    assert nd == 0 and la == lb


if __name__ == '__main__':
    main()
