"""Which Python lines launch the torch-side kernels of one smoke (or, with the argument `burgers`, Burgers batch-16) training step (copies, fills, adds): a TorchDispatchMode over one step
with the autograd engine on the calling thread, every aten op that launches a kernel grouped by its innermost wdno_amd frames
(run on the GPU box)."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from wdno_amd import _lib
from wdno_amd.trainer import TrainStep, multistep_lr
_lib.load()
dev = torch.device('cuda', 0)
burgers = 'burgers' in sys.argv[1:]
dif = bench.build_burgers(dev) if burgers else bench.build_model(dev, 8)
ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
batch = (torch.randn(16, 9, 64, 64) * 0.5).to(dev) if burgers else (torch.randn(8, 24, 42, 40, 40) * 0.5).to(dev)
for _ in range(3):
    ts.step(batch)
torch.cuda.synchronize()
SKIP = ('aten.view', 'aten.reshape', 'aten._unsafe_view', 'aten.detach', 'aten.alias', 'aten.empty', 'aten.as_strided', 'aten.slice', 'aten.select',
        'aten.permute', 'aten.transpose', 'aten.t.', 'aten.expand', 'aten.unsqueeze', 'aten.squeeze', 'aten.empty_like', 'aten.new_empty',
        'aten.is_', 'aten.sym_', 'aten.stride', 'aten.size', 'aten.narrow', 'aten.unbind', 'aten.split', 'aten._local_scalar_dense', 'aten.item',
        'aten.lift_fresh', 'aten.set_', 'aten.resize_', 'aten.record_stream', 'aten.empty_strided', 'aten.view_as', 'aten._reshape_alias')
counts = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            ts_ = [a for a in args if torch.is_tensor(a)]
            if any(t.is_cuda for t in ts_) or 'device' in (kwargs or {}):
                fr = [f for f in traceback.extract_stack()[:-1] if ('wdno_amd' in f.filename or 'bench.py' in f.filename)]
                where = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in fr[-3:][::-1]) or 'autograd engine'
                shape = tuple(ts_[0].shape) if ts_ else ()
                counts[(name, shape if len(shape) < 3 and (not shape or max(shape) < 4096) else 'big', where)] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)
with Log():
    ts.step(batch)
torch.cuda.synchronize()
tot = 0
for k, v in counts.most_common(60):
    print(f'{v:4d} x {k[0]:28s} {str(k[1]):16s} {k[2]}')
    tot += v
print('total logged ops:', sum(counts.values()))
