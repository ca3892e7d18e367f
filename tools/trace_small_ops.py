"""Which Python lines of the product path issue the small torch launches of a training step (copies, fills, adds): one eager step of the smoke
model with the torch entry points that launch such kernels wrapped to record their caller.   python tools/trace_small_ops.py"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd.trainer import TrainStep, multistep_lr

dev = torch.device('cuda', 0)
dif = bench.build_model(dev, 8)
ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
x = (torch.randn(8, 24, 42, 40, 40) * 0.5).to(dev)
for _ in range(3):
    ts.step(x)
torch.cuda.synchronize()
sites = collections.Counter()
active = [False]


def where():
    for f in reversed(traceback.extract_stack(limit=14)[:-2]):
        if 'tools/trace_small_ops' in f.filename:
            continue
        if any(k in f.filename for k in ('wdno_amd', 'third_party', 'video_diffusion', 'ddpm')):
            return f'{os.path.relpath(f.filename, ROOT)}:{f.lineno} {f.line[:70] if f.line else ""}'
    return 'outside'


def wrap(owner, name, label):
    real = getattr(owner, name)

    def w(*a, **k):
        if active[0]:
            t = next((v for v in a if torch.is_tensor(v)), None)
            if t is None or t.is_cuda or name in ('zeros', 'full', 'ones', 'zeros_like', 'tensor', 'cat', 'stack'):
                sites[(label, where())] += 1
        return real(*a, **k)
    setattr(owner, name, w)


for nm in ('zero_', 'fill_', 'copy_', 'clone', 'contiguous', 'new_zeros', 'to', 'float', 'add', 'add_', 'mul', 'mul_', 'sum', '__add__', '__mul__', '__getitem__', 'sqrt', 'expand', 'repeat'):
    wrap(torch.Tensor, nm, 'Tensor.' + nm)
for nm in ('zeros', 'full', 'ones', 'zeros_like', 'cat', 'stack', 'tensor', 'where', 'randn', 'randn_like', 'randint', 'rand'):
    wrap(torch, nm, 'torch.' + nm)
active[0] = True
ts.step(x)
torch.cuda.synchronize()
active[0] = False
for (name, frame), n in sites.most_common(70):
    print(f'{n:4d}  {name:20s} {frame}')
