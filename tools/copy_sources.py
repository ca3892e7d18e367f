"""Where do the device-to-device copies (aten::copy_, aten::clone, aten::contiguous) of a train step come from?"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import _lib
from wdno_amd.trainer import TrainStep, multistep_lr
_lib.load()
dev = torch.device('cuda', 0)
dif = bench.build_model(dev, 8)
ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
batch = (torch.randn(8, 24, 42, 40, 40) * 0.5).to(dev)
for _ in range(3):
    ts.step(batch)
torch.cuda.synchronize()
counts = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if self.is_cuda and (name != 'contiguous' or not self.is_contiguous()):
            fr = [x for x in traceback.extract_stack(limit=8)[:-1] if 'wdno_amd' in x.filename or 'bench' in x.filename]
            where = ' <- '.join(f'{os.path.basename(x.filename)}:{x.lineno}' for x in fr[-3:][::-1])
            counts[(name, tuple(self.shape) if self.numel() < 1e5 else 'big', where)] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
for n in ('copy_', 'clone', 'contiguous', 'to'):
    wrap(n)
ts.step(batch)
torch.cuda.synchronize()
for k, v in counts.most_common(25):
    print(v, k)
