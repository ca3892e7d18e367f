"""Which tensors does one smoke training step still split into fp16 planes with a pass of its own (ops.split_f16 / split_f16_colsum),
by call site and shape?"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import _lib, ops
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
    out = []
    for fr in reversed(traceback.extract_stack(limit=18)[:-2]):
        if 'wdno_amd' in fr.filename and not fr.filename.endswith('ops.py'):
            out.append(f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}')
            if len(out) == 2:
                break
    return ' < '.join(out) if out else 'backward'


for name in ('split_f16', 'split_f16_colsum'):
    orig = getattr(ops, name)

    def mk(orig, name):
        def f(x2d, *a, **k):
            if on[0]:
                counts[(name, tuple(x2d.shape), site())] += 1
            return orig(x2d, *a, **k)
        return f
    setattr(ops, name, mk(orig, name))
on[0] = True
ts.step(x)
torch.cuda.synchronize()
on[0] = False
tot = 0
for k, v in counts.most_common(40):
    mb = k[1][0] * k[1][1] * 8 / 1e6 * v
    tot += mb
    print(v, k, f'{mb:.0f} MB moved')
print('total MB moved by split passes per step:', round(tot))
