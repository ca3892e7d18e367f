"""world_size = 2 gloo test of the data-parallel exchange: a flat gradient buffer per rank, one
all-reduce(sum) and the 1/world scale give every rank the mean gradient -- exactly what TrainStep does over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from wdno_amd.trainer import FlatBuffers, allreduce_sum_
    torch.manual_seed(0)                                       # identical replicas
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    buf = FlatBuffers(model.parameters())
    g = torch.Generator().manual_seed(100 + rank)              # each rank sees its own shard of the global batch
    x = torch.randn(4, 6, generator=g)
    y = torch.randn(4, 3, generator=g)
    buf.zero_grad()
    ((model(x) - y) ** 2).mean().backward()
    buf.gather_grads()
    assert all(p.grad.data_ptr() == buf.flat_grad[o:o + 1].data_ptr() for p, (o, n) in zip(buf.params, buf._spans()))
    allreduce_sum_(buf.flat_grad, world)
    mean_grad = buf.flat_grad / world
    out[rank] = mean_grad.clone()
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0], out[1])
    # single-process reference: mean over the two shards of the per-shard mean losses
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    tot = 0
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(4, 6, generator=g)
        y = torch.randn(4, 3, generator=g)
        tot = tot + ((model(x) - y) ** 2).mean() / world
    tot.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(out[0], ref, rtol=1e-6, atol=1e-7)


def _worker_overlap(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from wdno_amd.trainer import FlatBuffers, OverlappedAllReduce, allreduce_sum_
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3),
                                torch.nn.Linear(3, 3))
    model[5].weight.requires_grad_(True)
    unused = torch.nn.Parameter(torch.ones(7))                  # never reached by backward: its bucket is launched by finish()
    buf = FlatBuffers(list(model.parameters()) + [unused])
    ar = OverlappedAllReduce(buf, n_buckets=3)
    assert len(ar.bounds) >= 2 and ar.bounds[0][0] == 0 and ar.bounds[-1][1] == buf.numel
    assert all(a[1] == b[0] for a, b in zip(ar.bounds, ar.bounds[1:]))          # contiguous, parameter-aligned spans
    g = torch.Generator().manual_seed(200 + rank)
    res = []
    for it in range(2):                                         # two steps: the hooks must re-arm
        x = torch.randn(4, 6, generator=g)
        y = torch.randn(4, 3, generator=g)
        buf.zero_grad()
        ar.begin()
        ((model(x) - y) ** 2).mean().backward()
        ar.finish()
        over = buf.flat_grad.clone()
        buf.zero_grad()                                         # the same step with one all-reduce after backward
        ((model(x) - y) ** 2).mean().backward()
        buf.gather_grads()
        allreduce_sum_(buf.flat_grad, world)
        res.append((over, buf.flat_grad.clone()))
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucket_allreduce_world2():
    """OverlappedAllReduce (buckets launched from gradient hooks during backward) gives the same flat gradient as the single
    all-reduce after backward, including a bucket holding a parameter that receives no gradient."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_overlap, args=(world, port, out), nprocs=world, join=True)
    for rank in range(world):
        for over, plain in out[rank]:
            assert torch.equal(over, plain)
    assert torch.equal(out[0][0][0], out[1][0][0])
