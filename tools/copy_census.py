"""Which Python lines issue the torch copy / fill / add kernels of one sampling step and one training step (torch profiler, with stacks)."""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from wdno_amd import diffusion_core as K
from wdno_amd.trainer import TrainStep

dev = 'cuda'
dif = bench.build_model(dev)
shape = (8, 24, 42, 40, 40)
x = torch.randn(shape, device=dev)
init = torch.randn(8, 24, 40, 40, device=dev)
control = torch.randn(8, 24, 16, 40, 40, device=dev)
desc = dif._desc(shape, dif.padded_shape)
src = dif._condition_source(shape, dev, init, control, None)


def sample_step():
    global x
    with torch.no_grad():
        x, _ = dif.p_sample(shape, x, 500)
        x = K.apply_cond(x, src, desc)


ts = TrainStep(dif, lr=1e-4)
xt = torch.randn(shape, device=dev)


def report(fn, title):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        fn()
        torch.cuda.synchronize()
    agg = collections.Counter()
    for ev in prof.events():
        if ev.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::cat', 'aten::normal_', 'aten::randn',
                       'aten::clone', 'aten::contiguous', 'aten::zeros', 'aten::sum', 'aten::sqrt', 'aten::div', 'aten::sub'):
            st = [s for s in ev.stack if 'wdno_amd' in s or 'bench.py' in s]
            shapes = str(ev.input_shapes)[:60]
            agg[(ev.name, shapes, ' | '.join(s.split('/')[-1] for s in st[:2]))] += 1
    print('==', title)
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1])[:45]:
        print(f'{c:3d} {k[0]:16s} {k[1]:62s} {k[2]}')


report(sample_step, 'sampling step')
report(lambda: ts.step(xt), 'training step')
