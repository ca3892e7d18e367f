import os, sys, torch
sys.path.insert(0, '/root/repo')
import bench
from wdno_amd import ops
from wdno_amd.trainer import TrainStep
dev = 'cuda'
dif = bench.build_burgers(dev)
ts = TrainStep(dif, lr=1e-4)
x = torch.randn(16, 9, 64, 64, device=dev)
for _ in range(4):
    ts.step(x)
torch.cuda.synchronize()
print('ok')
