"""Where the standalone passes of one smoke training step come from: every ops.split_f16 / amax call and every torch copy / add /
fill kernel, grouped by the model-code frame that asked for it (run on the GPU box)."""
import collections
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import ops
from wdno_amd.trainer import TrainStep

dev = 'cuda'
dif = bench.build_model(dev)
ts = TrainStep(dif, lr=1e-4)
x = torch.randn(8, 24, 42, 40, 40, device=dev)
for _ in range(2):
    ts.step(x)
torch.cuda.synchronize()


def where():
    fr = [f for f in traceback.extract_stack()[:-2] if 'wdno_amd' in f.filename]
    own = [f for f in fr if not f.filename.endswith('ops.py')]
    inner = [f for f in fr if f.filename.endswith('ops.py')]
    a = own[-1] if own else None
    b = inner[-1] if inner else None
    s = f'{os.path.basename(a.filename)}:{a.lineno} {a.line}' if a else '?'
    return s + (f'   <- ops.py:{b.lineno} ({b.name})' if b else '')


counts = collections.Counter()
phase = ['fwd']
for name in ('split_f16', 'split_f16_colsum', 'amax_record', 'tensor_amax'):
    if not hasattr(ops, name):
        continue
    orig = getattr(ops, name)

    def wrap(*a, _o=orig, _n=name, **k):
        shape = tuple(a[0].shape) if a and hasattr(a[0], 'shape') else ()
        counts[(_n, shape, where())] += 1
        return _o(*a, **k)
    setattr(ops, name, wrap)
ts.step(x)
torch.cuda.synchronize()
print('== standalone split / amax passes of one step')
for (n, shape, w), c in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f'{c:3d} x {n:18s} {str(shape):18s} {w}')

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step(x)
    torch.cuda.synchronize()
agg = collections.Counter()
tot = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::mul', 'aten::div', 'aten::cat', 'aten::clone', 'aten::contiguous', 'aten::sub', 'aten::neg', 'aten::mul_', 'aten::to') and ev.device_time_total > 0:
        st = [s for s in ev.stack if 'wdno_amd' in s]
        key = (ev.name, st[0] if st else '?', st[1] if len(st) > 1 else '')
        agg[key] += 1
        tot[key] += ev.device_time_total
print('== torch kernels of one step by origin (count, device us)')
for k, c in sorted(agg.items(), key=lambda kv: -tot[kv[0]])[:40]:
    print(f'{c:3d} {tot[k]:8.1f} us  {k[0]:12s} {k[1]}  <-  {k[2]}')
