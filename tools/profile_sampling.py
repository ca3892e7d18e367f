"""Eager DDPM sampling steps of the smoke base model (batch 8, or argv[2]) for a kernel trace (rocprofv3 --kernel-trace -- python tools/profile_sampling.py [steps [batch]])."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import diffusion_core as K

dev = 'cuda'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dif = bench.build_model(dev, B)
shape = (B, 24, 42, 40, 40)
x = torch.randn(shape, device=dev)
init = torch.randn(B, 24, 40, 40, device=dev)
control = torch.randn(B, 24, 16, 40, 40, device=dev)
desc = dif._desc(shape, dif.padded_shape)
src = dif._condition_source(shape, dev, init, control, None)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with torch.no_grad():
    for i in range(n):
        x, _ = dif.p_sample(shape, x, 500 - i)
        x = K.apply_cond(x, src, desc)
torch.cuda.synchronize()
print('done', n)
