"""CPU, world_size 2, gloo: the data-parallel gradient reducer (bucketing from the tail of the backward
order, async all-reduce per bucket, mean, arena-contiguous and scattered gradients)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, arena, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rho_diffusion_amd.parallel import GradBucketReducer, broadcast_parameters
    torch.manual_seed(1234 + rank)                      # different initial weights per rank
    model = nn.Sequential(nn.Linear(64, 96), nn.Linear(96, 200), nn.Linear(200, 8))
    broadcast_parameters(model)
    params = list(model.parameters())
    if arena:                                           # gradients adjacent in memory, as in the HipAdamW arena
        flat = torch.zeros(sum(p.numel() for p in params))
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
    red = GradBucketReducer(params, bucket_bytes=40_000)
    assert len(red.buckets) >= 2
    for step in range(2):
        g = torch.Generator().manual_seed(10 * step + rank)
        for p in params:
            val = torch.randn(p.shape, generator=g)
            if p.grad is None:
                p.grad = val
            else:
                p.grad.copy_(val)
        # backward finalises gradients from the tail to the head, in groups
        red.on_ready(params[4:])
        red.on_ready(params[2:4])
        red.on_ready(params[:2])
        red.finish()
        # expected: mean over ranks of the same generator streams
        exp = []
        for r in range(world):
            gg = torch.Generator().manual_seed(10 * step + r)
            exp.append([torch.randn(p.shape, generator=gg) for p in params])
        for i, p in enumerate(params):
            want = sum(e[i] for e in exp) / world
            assert torch.allclose(p.grad, want, atol=1e-6), (rank, step, i)
    w0 = [p.detach().clone() for p in params]
    gathered = [None] * world
    dist.all_gather_object(gathered, [float(w.sum()) for w in w0])
    assert gathered[0] == gathered[1]                  # broadcast_parameters made the replicas identical
    q.put((rank, "ok"))
    dist.destroy_process_group()


@pytest.mark.parametrize("arena", [False, True])
def test_grad_bucket_reducer_world2(arena):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, arena, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


def test_reducer_is_noop_without_process_group():
    from rho_diffusion_amd.parallel import GradBucketReducer
    m = nn.Linear(4, 4)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    red = GradBucketReducer(list(m.parameters()))
    red.on_ready(list(m.parameters()))
    red.finish()
    assert all(float(p.grad.mean()) == 1.0 for p in m.parameters())
