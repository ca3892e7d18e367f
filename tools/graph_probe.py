"""Does capturing forward + backward of the smoke training step in a HIP graph pay? (torch.cuda.graph; ops launch on the
capture stream through ctypes.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import _lib, ops
_lib.load()
dev = torch.device('cuda', 0)
dif = bench.build_model(dev, 8)
batch = (torch.randn(8, 24, 42, 40, 40) * 0.5).to(dev)
params = [p for p in dif.parameters() if p.requires_grad]

# under capture the amax slots must be zeroed by captured kernels
_orig_slot = ops._amax_slot
def slot(device, n=1):
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(ops.AMAX_FLOATS, device=device, dtype=torch.float32)
    return _orig_slot(device) if n == 1 else _orig_slot(device, n)
ops._amax_slot = slot

def fwd_bwd():
    for p in params:
        p.grad = None
    loss = dif(batch)
    loss.backward()
    return loss

for _ in range(3):
    fwd_bwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    fwd_bwd()
torch.cuda.synchronize()
print('eager fwd+bwd ms:', (time.perf_counter() - t0) / 5 * 1e3)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        fwd_bwd()
torch.cuda.current_stream().wait_stream(s)
ops.bump_weight_epoch()
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    static_loss = fwd_bwd()
torch.cuda.synchronize()
print('capture s:', time.perf_counter() - t0)
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print('graph replay fwd+bwd ms:', (time.perf_counter() - t0) / 5 * 1e3, 'loss', float(static_loss))
