"""Which Python call sites issue the small torch kernels (copies, fills, adds, cats) of one smoke training step?
python tools/trace_glue.py  ->  counts per (aten op, innermost wdno_amd frame)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import _lib
from wdno_amd.trainer import TrainStep, multistep_lr

dev = torch.device('cuda', 0)
_lib.load()
dif = bench.build_model(dev, 8)
ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
x = (torch.randn(8, 24, 42, 40, 40) * 0.5).to(dev)
ts.step(x); ts.step(x)
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step(x)
    torch.cuda.synchronize()
counts = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::fill_', 'aten::add', 'aten::add_', 'aten::cat', 'aten::zero_', 'aten::mul', 'aten::clone', 'aten::contiguous',
                   'aten::zeros', 'aten::zeros_like', 'aten::sum', 'aten::div', 'aten::sqrt') and ev.device_type == torch.autograd.DeviceType.CPU:
        site = next((s for s in ev.stack if 'wdno_amd' in s or 'bench.py' in s), ev.stack[0] if ev.stack else '?')
        counts[(ev.name, site.strip()[-110:])] += 1
for k, v in counts.most_common(45):
    print(v, k)
