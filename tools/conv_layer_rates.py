"""Per-layer rates of the convolution launches of one training step (ops.PROFILE: HIP events around every launch): kernel family, algorithmic
GFLOP per launch, launches, mean time, TFLOP/s -- layers far below their family's rate are candidates for another tile shape.
usage: python tools/conv_layer_rates.py [smoke|burgers|burgers80]"""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import ops
from wdno_amd.trainer import TrainStep

which = sys.argv[1] if len(sys.argv) > 1 else 'smoke'
dev = torch.device('cuda', 0)
if which == 'smoke':
    dif, shape = bench.build_model(dev, 8), (8, 24, 42, 40, 40)
elif which == 'burgers':
    dif, shape = bench.build_burgers(dev), (16, 9, 64, 64)
else:
    dif, shape = bench.build_burgers(dev, (80, 64)), (16, 9, 80, 64)
ts = TrainStep(dif, lr=1e-3, use_ema=False)
x = torch.randn(shape, device=dev) * 0.5
for _ in range(3):
    ts.step(x)
ops.PROFILE = {}
ts.step(x)
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
rows = collections.defaultdict(list)
for name, evs in prof.items():
    for e0, e1, fl in evs:
        rows[(name, round(fl / 1e9, 2))].append(e0.elapsed_time(e1))
tot = sum(sum(v) for v in rows.values())
print(f'{which}: {tot:.2f} ms in convolution launches')
for (name, gf), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    ms = sum(v) / len(v)
    print(f'{sum(v):7.3f} ms  {len(v):3d} x {ms * 1e3:7.1f} us  {gf:8.2f} GFLOP  {gf / ms:7.1f} TFLOP/s  {name}')
