import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from wdno_amd import ops
from wdno_amd.trainer import TrainStep, multistep_lr, cosine_annealing_lr
dev = torch.device('cuda', 0)
out = {}
CASES = (('smoke', 'f16x3', lambda: bench.build_model(dev, 8), (8, 24, 42, 40, 40)), ('burgers', 'f16x3', lambda: bench.build_burgers(dev), (16, 9, 64, 64)),
         ('burgers80', 'f16x3', lambda: bench.build_burgers(dev, (80, 64)), (16, 9, 80, 64)),
         # BASELINE configs[1] at its own batch and the smoke step on the single-product kernels (bench leg smoke_bf16)
         ('burgers_bf16_b256', 'bf16', lambda: bench.build_burgers(dev), (256, 9, 64, 64)), ('smoke_bf16', 'bf16', lambda: bench.build_model(dev, 8), (8, 24, 42, 40, 40)),
         ('smoke34', 'f16x3', lambda: bench.build_model(dev, 8, channels=34), (8, 24, 34, 40, 40)))
for name, math, build, shape in CASES:
    ops.CONV_MATH = math
    dif = build()
    ts = TrainStep(dif, lr=1e-3, use_ema=False)
    x = torch.randn(shape, device=dev) * 0.5
    ts.step(x); ts.step(x)
    ops.PROFILE = {}
    ts.step(x)
    torch.cuda.synchronize()
    out[name] = {k: len(v) for k, v in sorted(ops.PROFILE.items())}
    ops.PROFILE = None
    del ts, dif
    ops.drop_weight_caches()
ops.CONV_MATH = 'f16x3'
print(json.dumps(out, indent=1))
