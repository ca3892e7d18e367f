"""Which packed-weight operands miss the two-launch batch refresh (ops._refresh_weight_plans) in a smoke train step?
Prints (shape, stride, kind, contiguous) of every operand that was repacked on its own after the third step."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from wdno_amd import _lib, ops  # noqa: E402
from wdno_amd.trainer import TrainStep, multistep_lr  # noqa: E402

_lib.load()
dev = torch.device('cuda:0')
dif = bench.build_model(dev, 8)
ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
x = (torch.randn(8, 24, 42, 40, 40) * 0.5).to(dev)
lib = ops._lib_()
real = lib.wdno_pack_split_weight
count = collections.Counter()
step = [0]
old = ops.split_weight


def traced(w, kind, cp8, kp, pack=None):
    key = (w.data_ptr(), kind, cp8, kp, tuple(w.shape), tuple(w.stride()), ops._lp())
    had = key in ops._wplans
    n0 = calls[0]
    r = old(w, kind, cp8, kp, pack)
    if calls[0] != n0 and step[0] >= 3:
        count[(tuple(w.shape), tuple(w.stride()), kind, bool(w.is_contiguous()), had)] += 1
    return r


calls = [0]


class Proxy:
    def __getattr__(self, name):
        fn = getattr(lib, name)
        if name == 'wdno_pack_split_weight':
            def wrapped(*a):
                calls[0] += 1
                return fn(*a)
            return wrapped
        return fn


ops.split_weight = traced
ops._lib_ = lambda: Proxy()
import traceback
where = collections.Counter()
_P = Proxy


class Proxy2(_P):
    def __getattr__(self, name):
        fn = getattr(lib, name)
        if name == 'wdno_pack_split_weight':
            def wrapped(*a):
                calls[0] += 1
                if step[0] >= 3:
                    fr = traceback.extract_stack(limit=6)
                    where[' <- '.join(f'{f.name}:{f.lineno}' for f in reversed(fr[:-1]))] += 1
                return fn(*a)
            return wrapped
        return fn


ops._lib_ = lambda: Proxy2()
for i in range(5):
    step[0] = i
    c0 = calls[0]
    ts.step(x)
    print('step', i, 'single packs', calls[0] - c0)
for k, v in where.most_common(8):
    print(v, k)
torch.cuda.synchronize()
for k, v in sorted(count.items(), key=lambda kv: -kv[1]):
    print(v, k)
