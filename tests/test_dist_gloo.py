"""world_size-2 `gloo` test of the data-parallel gradient exchange (cocosnet_amd/dist.py) on CPU.
The same code runs over RCCL (`nccl`) on the GPU box; here it proves the bucketing / averaging /
sharding logic: after all_reduce_ every rank holds the mean of the per-rank gradients, which equals
the gradient of the mean loss over the GLOBAL batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(5, 7, 1), torch.nn.PReLU(), torch.nn.Conv2d(7, 3, 3, padding=1))


def _data(global_batch=4):
    g = torch.Generator().manual_seed(1)
    return torch.randn(global_batch, 5, 6, 6, generator=g), torch.randn(global_batch, 3, 6, 6, generator=g)


def _worker(rank, world, port, bucket_bytes, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from cocosnet_amd import dist as cdist
    r, lr, w = cdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    model = _model()
    x, y = _data()
    lo, hi = cdist.shard_batch(x.shape[0], rank, world)
    loss = ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean()
    loss.backward()
    buckets = cdist.GradBuckets(model.parameters(), bucket_bytes=bucket_bytes)
    buckets.all_reduce_()
    ret[rank] = [p.grad.clone() for p in model.parameters()]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [64, 1 << 20])   # many tiny buckets / one bucket
def test_gradient_allreduce_equals_global_batch_gradient(bucket_bytes):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, bucket_bytes, ret), nprocs=world, join=True)
    model = _model()
    x, y = _data()
    ((model(x) - y) ** 2).mean().backward()
    ref = [p.grad for p in model.parameters()]
    for rank in range(world):
        for a, b in zip(ret[rank], ref):
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-5)


def test_bucket_layout_and_sharding():
    from cocosnet_amd import dist as cdist
    model = _model()
    b = cdist.GradBuckets(model.parameters(), bucket_bytes=64)
    assert sum(len(x) for x in b.buckets) == len(list(model.parameters()))
    assert b.buckets[0][0] is list(model.parameters())[-1]          # reverse (gradient-ready) order
    assert b.nbytes() == sum(p.numel() for p in model.parameters()) * 4
    assert cdist.shard_batch(64, 3, 8) == (24, 32)
    with pytest.raises(ValueError):
        cdist.shard_batch(10, 0, 4)
    assert cdist.init_from_env() == (0, 0, 1) or os.environ.get("WORLD_SIZE", "1") != "1"
