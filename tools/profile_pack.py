"""The smoke fields -> state pipeline (3-D DWT launch + packing launch) twenty times, for a kernel trace:
rocprofv3 --kernel-trace --stats -- python tools/profile_pack.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
bench._trees()
from ddpm.data_2d import _RESCALERS

dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
resc = torch.tensor(_RESCALERS['bior1.3'], dtype=torch.float32, device=dev).reshape(1, 42, 1, 1)
fields = torch.randn(B, 5, 32, 64, 64, device=dev)
curve = torch.rand(B, 32, device=dev)
for _ in range(20):
    st = bench.smoke_fields_to_state(fields, resc, curve)
torch.cuda.synchronize()
print('done', tuple(st.shape))
