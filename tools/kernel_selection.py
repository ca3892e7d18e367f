import sys, json, torch
sys.path.insert(0, '/root/repo')
import bench
from wdno_amd import ops
from wdno_amd.trainer import TrainStep, multistep_lr, cosine_annealing_lr
dev = torch.device('cuda', 0)
out = {}
for name, build, shape in (('smoke', lambda: bench.build_model(dev, 8), (8, 24, 42, 40, 40)), ('burgers', lambda: bench.build_burgers(dev), (16, 9, 64, 64)), ('burgers80', lambda: bench.build_burgers(dev, (80, 64)), (16, 9, 80, 64))):
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
print(json.dumps(out, indent=1))
