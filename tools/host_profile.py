"""cProfile of the host side of the train step (enqueue only; the GPU is drained before and after each profiled step)."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import _lib, ops
from wdno_amd.trainer import TrainStep, multistep_lr
_lib.load()
dev = torch.device('cuda', 0)
dif = bench.build_model(dev, 8)
ts = TrainStep(dif, lr=1e-3, betas=(0.9, 0.99), max_grad_norm=1.0, lr_schedule=multistep_lr, use_ema=True)
batch = (torch.randn(8, 24, 42, 40, 40) * 0.5).to(dev)
for _ in range(3):
    ts.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    ts.step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f'enqueue of 4 steps back to back: {(t1 - t0) * 250:.1f} ms per step (the GPU paces this once its queue is full)')
pr = cProfile.Profile()
for _ in range(4):
    torch.cuda.synchronize()
    pr.enable()
    ts.step(batch)
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(32)
