"""Is the training step GPU-bound or host(launch)-bound? Times (a) the host-side enqueue of one step without waiting and
(b) the wall time per step, over a few steps, and counts launches."""
import os, sys, time
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
enq, wall = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
print('host enqueue ms per step:', [round(v, 1) for v in enq])
print('wall ms per step (single step, sync before/after):', [round(v, 1) for v in wall])

# ---- the same with the step captured in one HIP graph (TrainStep.capture)
ts.capture(batch, warmup=1)
for _ in range(3):
    ts.step(batch)
torch.cuda.synchronize()
enq, wall = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts.step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
print('graph replay: host enqueue ms per step:', [round(v, 1) for v in enq])
print('graph replay: wall ms per step (single step, sync before/after):', [round(v, 1) for v in wall])
t0 = time.perf_counter()
for _ in range(30):
    ts.step(batch)
torch.cuda.synchronize()
print('graph replay: ms per step back to back (30 steps):', round((time.perf_counter() - t0) / 30 * 1e3, 2))
