"""aten::copy_ / add_ / fill_ / zero_ of one smoke training step with their input shapes and the autograd node they ran under."""
import collections, os, sys
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
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True) as prof:
    ts.step(x)
    torch.cuda.synchronize()
evs = [e for e in prof.events()]
counts = collections.Counter()
for e in evs:
    if e.name in ('aten::copy_', 'aten::add_', 'aten::add', 'aten::fill_', 'aten::zero_', 'aten::clone', 'aten::cat', 'aten::mul', 'aten::sum'):
        par = e.cpu_parent
        chain = []
        while par is not None and len(chain) < 3:
            chain.append(par.name)
            par = par.cpu_parent
        counts[(e.name, str(e.input_shapes)[:70], ' < '.join(chain)[:110])] += 1
for k, v in counts.most_common(45):
    print(v, k)
