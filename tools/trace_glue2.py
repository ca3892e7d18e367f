"""Python call sites of Tensor.clone / contiguous (when it copies) / copy_ / zeros_like / torch.zeros / add_ in one smoke training step
(the torch profiler of this build records no Python stacks for tools/trace_glue.py)."""
import collections
import os
import sys
import traceback

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
counts = collections.Counter()
on = [False]


def site():
    for fr in reversed(traceback.extract_stack(limit=14)[:-2]):
        if 'wdno_amd' in fr.filename and 'trace_glue' not in fr.filename:
            return f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}'
    return '?'


def wrap(cls, name, cond=lambda self, *a, **k: True):
    orig = getattr(cls, name)

    def f(self, *a, **k):
        if on[0] and cond(self, *a, **k):
            counts[(name, site(), tuple(self.shape) if hasattr(self, 'shape') else None)] += 1
        return orig(self, *a, **k)
    setattr(cls, name, f)


wrap(torch.Tensor, 'clone')
wrap(torch.Tensor, 'contiguous', lambda self, *a, **k: not self.is_contiguous())
wrap(torch.Tensor, 'copy_')
wrap(torch.Tensor, 'zero_')
wrap(torch.Tensor, 'fill_')
wrap(torch.Tensor, 'add_')
wrap(torch.Tensor, 'new_zeros')
wrap(torch.Tensor, '__getitem__', lambda self, *a, **k: self.is_cuda and self.dim() >= 1 and self.numel() > 1 and False)
for fn in ('zeros', 'zeros_like', 'cat', 'stack'):
    o = getattr(torch, fn)

    def mk(o, fn):
        def f(*a, **k):
            if on[0]:
                counts[(fn, site(), None)] += 1
            return o(*a, **k)
        return f
    setattr(torch, fn, mk(o, fn))
on[0] = True
ts.step(x)
torch.cuda.synchronize()
on[0] = False
for k, v in counts.most_common(40):
    print(v, k)
