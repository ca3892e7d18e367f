"""A few Burgers training steps (batch 16, [16,9,64,64]) for a kernel trace: rocprofv3 --kernel-trace -- python tools/profile_burgers_step.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd.trainer import TrainStep
dev = 'cuda'
dif = bench.build_burgers(dev)
ts = TrainStep(dif, lr=1e-4)
x = torch.randn(16, 9, 64, 64, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    ts.step(x)
torch.cuda.synchronize()
print('ok')
