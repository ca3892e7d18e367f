"""Debugging aid (GPU box): the overlapped bucket exchange over a one-rank RCCL group vs the plain path, gradient by gradient."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29671', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
import test_gpu_distributed as T  # noqa: E402

T._trees()
import torch.distributed as dist  # noqa: E402
from wdno_amd.trainer import TrainStep  # noqa: E402

torch.cuda.set_device(0)
x0, t, noise = T._batches()[0]


def grads(sync_in_hooks=False):
    dif = T._model(5).to('cuda')
    ts = TrainStep(dif, lr=1e-3, max_grad_norm=1.0, use_ema=False)
    ts.opt.zero_grad()
    loss = dif.p_losses(x0.cuda(), t.cuda(), noise=noise.cuda())
    ts._backward_and_exchange(loss)
    torch.cuda.synchronize()
    return ts, ts.opt.buf.flat_grad.detach().cpu().clone()


ts0, g0 = grads()
dist.init_process_group('nccl', rank=0, world_size=1)
os.environ['WDNO_DP_FORCE_EXCHANGE'] = '1'
os.environ['WDNO_DP_OVERLAP'] = '1'
ts1, g1 = grads()
print('overlap', ts1.overlap is not None, 'bounds', ts1.overlap.bounds)
names = [n for n, p in ts1.model.named_parameters() if p.requires_grad]
bad = 0
for (o, n), name, b in zip(ts1.opt.buf.span_list, names, ts1.overlap.bucket_of):
    a, c = g0[o:o + n], g1[o:o + n]
    if not torch.equal(a, c):
        bad += 1
        if bad < 15:
            print('differs: bucket', b, name, n, 'max abs diff', (a - c).abs().max().item(), 'ref max', a.abs().max().item(), 'got all zero:', bool((c == 0).all()))
print('params differing:', bad, 'of', len(names))
dist.destroy_process_group()
