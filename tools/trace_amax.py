"""Who asks for a stand-alone amax sweep / split in one smoke training step (tensors whose producer left no amax record)?
Prints (caller chain, shape) counts. python tools/trace_amax.py"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import ops, _lib
from wdno_amd.trainer import TrainStep, multistep_lr

dev = torch.device('cuda', 0)
_lib.load()
dif = bench.build_model(dev, 8)
ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
x = (torch.randn(8, 24, 42, 40, 40) * 0.5).to(dev)
ts.step(x); ts.step(x)
counts = collections.Counter()
orig_amax, orig_split = ops.tensor_amax, ops.split_f16


def chain(skip=2, depth=5):
    fr = traceback.extract_stack()[:-skip]
    return ' < '.join(f'{f.name}:{f.lineno}' for f in reversed(fr[-depth:]))


def amax(t):
    counts[('amax', tuple(t.shape), chain())] += 1
    return orig_amax(t)


def split(x2d, amax=None):
    counts[('split', tuple(x2d.shape), 'with record' if amax is not None else 'NO record', chain(depth=4))] += 1
    return orig_split(x2d, amax)


ops.tensor_amax, ops.split_f16 = amax, split
ts.step(x)
torch.cuda.synchronize()
for k, v in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(v, k)
