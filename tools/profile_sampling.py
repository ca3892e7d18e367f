"""A few eager DDPM sampling steps of the smoke model at the bench batch (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import _lib, diffusion_core as K

dev = torch.device('cuda', 0)
_lib.load()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dif = bench.build_model(dev, b)
shape = (b, 24, 42, 40, 40)
x = torch.randn(shape, device=dev)
init = torch.randn(b, 24, 40, 40, device=dev)
control = torch.randn(b, 24, 16, 40, 40, device=dev)
desc = dif._desc(shape, dif.padded_shape)
src = dif._condition_source(shape, dev, init, control, None)
with torch.no_grad():
    for i in range(6):
        x, _ = dif.p_sample(shape, x, 500 - i)
        x = K.apply_cond(x, src, desc)
torch.cuda.synchronize()
