"""-m gpu: the data-parallel trainer end to end on 2 ranks.  Both ranks share the single GPU of the test box, so
the process group uses gloo (which moves GPU tensors through the host); the code path (broadcast, arena-ordered
buckets, on_ready callbacks from the backward plan, averaged gradients, fused AdamW) is the one RCCL runs under
torchrun.  Checks: replicas stay bit-identical, and equal a single-process run on the rank-averaged gradients."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build():
    from torch import nn
    from helpers import UNET_CASES, det_state_dict
    from rho_diffusion_amd.diffusion import DDPM, LinearSchedule
    from rho_diffusion_amd.models import UNet
    kw, xshape, _ = UNET_CASES["tiny2d"]
    ddpm = DDPM(UNet, dict(kw, compute_dtype="fp32"), LinearSchedule(1000, 1e-3, 0.02), nn.MSELoss, opt_kwargs={"lr": 2e-4})
    ddpm.backbone.load_state_dict(det_state_dict(ddpm.backbone.state_dict(), "ddp"))
    return ddpm.to("cuda"), xshape


def _batch(rank, step, xshape):
    from helpers import det_normal, det_uniform
    x0 = det_uniform(xshape, f"ddp_x{rank}_{step}", 0.0, 1.0).to("cuda")
    eps = det_normal(xshape, f"ddp_e{rank}_{step}").to("cuda")
    t = torch.tensor([(97 * (rank + 1) + 31 * step + 13 * i) % 1000 for i in range(xshape[0])])
    return x0, eps, t


def _worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rho_diffusion_amd.trainer import DPTrainer
    ddpm, xshape = _build()
    if rank == 1:                      # replicas start different: the trainer must broadcast rank 0
        with torch.no_grad():
            for p_ in ddpm.parameters():
                p_.add_(0.01)
    trainer = DPTrainer(ddpm, scale_lr_by_sqrt_world=False, bucket_bytes=400_000)
    assert len(trainer.reducer.buckets) >= 3
    for step in range(2):
        x0, eps, t = _batch(rank, step, xshape)
        ddpm.noise = lambda data, e=eps: e
        ddpm.random_timesteps = lambda bs, tt=t: tt
        trainer.step(x0)
    flat = torch.cat([p_.detach().reshape(-1) for p_ in ddpm.backbone.parameters()]).cpu()
    q.put((rank, flat.numpy()))      # by value: a torch tensor travels as an fd owned by this (exiting) process
    dist.barrier()
    dist.destroy_process_group()


def _collect(procs, q, n, timeout=300):
    """n results from the workers' queue; a worker that dies fails the test at once, and no child outlives the test
    (a surviving rank would sit in a collective holding the GPU)."""
    import queue
    import time
    out = []
    try:
        deadline = time.time() + timeout
        while len(out) < n:
            try:
                out.append(q.get(timeout=1.0))
            except queue.Empty:
                dead = [p for p in procs if p.exitcode not in (None, 0)]
                assert not dead, f"worker exited with {[p.exitcode for p in dead]}"
                assert time.time() < deadline, "timed out waiting for the workers"
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        return out
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(10)
            if p.is_alive():
                p.kill()
                p.join(10)


def test_dp_trainer_two_ranks_match_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: torch.from_numpy(flat) for r, flat in _collect(procs, q, 2)}
    assert torch.equal(got[0], got[1])                     # replicas identical after 2 steps

    # single process: same two "rank" batches per step, gradients averaged by hand, same fused AdamW
    from rho_diffusion_amd.optim import HipAdamW
    ddpm, xshape = _build()
    ddpm.train()
    opt = HipAdamW(ddpm.parameters(), lr=2e-4, arena_order=ddpm.backbone.engine().param_order())
    for step in range(2):
        opt.zero_grad()
        for rank in range(2):
            x0, eps, t = _batch(rank, step, xshape)
            ddpm.noise = lambda data, e=eps: e
            ddpm.random_timesteps = lambda bs, tt=t: tt
            (ddpm.training_step(x0) * 0.5).backward()      # mean over the two ranks
        opt.step()
    ref = torch.cat([p_.detach().reshape(-1) for p_ in ddpm.backbone.parameters()]).cpu()
    err = float((got[0] - ref).norm() / ref.norm())
    assert err < 2e-5, err     # fp32 summation order only (collective sum, wgrad atomics)


def _rccl_single_rank(port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from rho_diffusion_amd.parallel import GradBucketReducer
    ps = [torch.nn.Parameter(torch.randn(1000 + 37 * i, device="cuda")) for i in range(6)]
    arena = torch.randn(sum(p.numel() for p in ps), device="cuda")
    off = 0
    for p in ps:                                        # gradients as adjacent views of one arena (HipAdamW layout)
        p.grad = arena[off:off + p.numel()].view_as(p)
        off += p.numel()
    before = arena.clone()
    red = GradBucketReducer(ps, bucket_bytes=8000)
    assert red.avg_in_collective and len(red.buckets) >= 2
    red.world = 2                                        # force the collective path on the 1-rank group (AVG over 1 rank = identity)
    for p in reversed(ps):
        red.on_ready([p])
    red.finish()
    t = torch.ones(5, device="cuda")
    dist.broadcast(t, src=0)
    torch.cuda.synchronize()
    ok = bool(torch.equal(arena, before)) and all(w is not None for w in [red])
    q.put(("ok" if ok else "mismatch", int(sum(1 for b in red.buckets))))
    dist.destroy_process_group()


def test_rccl_backend_bucketed_all_reduce_single_rank():
    """The RCCL ("nccl") code path of the gradient reducer on the one GPU of the test box: communicator creation,
    asynchronous ReduceOp.AVG all-reduce on flat arena views, wait, broadcast.  (Multi-rank numerics are covered by the
    gloo tests; the 8-GPU run belongs to the driver.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank, args=(_free_port(), q))
    p.start()
    (status, nb), = _collect([p], q, 1)
    assert status == "ok" and nb >= 2 and p.exitcode == 0


def test_all_reduce_is_issued_while_backward_is_still_running(monkeypatch):
    """'RCCL all-reduce overlapped with backward'.  With a recording stand-in for dist.all_reduce, the first bucket's
    collective must be issued BEFORE the last weight-gradient kernel of the backward plan is launched (i.e. from an on_ready
    mark inside run_backward, not from finish()), buckets must go out in tail-to-head order of the arena, every bucket exactly
    once, and all of them be flat arena views (zero-copy).  Also the bf16-bucket option: half the bytes, fp32 gradients restored."""
    import torch.distributed as dist
    from helpers import det_uniform
    from rho_diffusion_amd import parallel
    from rho_diffusion_amd.trainer import DPTrainer
    for comm_dtype in (torch.float32, torch.bfloat16):
        ddpm, xshape = _build()
        trainer = DPTrainer(ddpm, bucket_bytes=400_000, comm_dtype=comm_dtype)
        red = trainer.reducer
        assert len(red.buckets) >= 3
        x0 = det_uniform(xshape, "ovl_x", 0.0, 1.0).to("cuda")
        trainer.step(x0)                                       # builds the training plan (world = 1: no collectives yet)
        plan = ddpm.backbone.engine()._last_train_plan
        events = []

        class _Work:
            def wait(self):
                events.append(("wait",))

        def fake_all_reduce(t, op=None, group=None, async_op=False):
            events.append(("all_reduce", t.data_ptr(), t.numel(), t.dtype))
            return _Work()

        monkeypatch.setattr(parallel.dist, "all_reduce", fake_all_reduce)
        red.world = 2                                          # arm the collective path on the single process
        red.avg_in_collective = True
        for i, (fn, info) in enumerate(zip(list(plan.bwd), plan.bwd_info)):
            plan.bwd[i] = (lambda s, fn=fn, kind=info["kind"]: (events.append(("launch", kind)), fn(s))[1])
        ref = None
        if comm_dtype == torch.bfloat16:
            ref = trainer.opt.flat_grads[0]
        trainer.step(x0)
        red.world = 1
        monkeypatch.undo()
        ar = [i for i, e in enumerate(events) if e[0] == "all_reduce"]
        wg = [i for i, e in enumerate(events) if e == ("launch", "wgrad")]
        assert len(ar) == len(red.buckets) and len(wg) > 10
        assert ar[0] < wg[-1], "first bucket was not issued before the last wgrad launch"
        assert sum(1 for i in ar if i < wg[-1]) >= len(red.buckets) - 2          # all but the head buckets go out under backward
        if comm_dtype == torch.float32:
            arena = trainer.opt.flat_grads[0]
            ptrs = [events[i][1] for i in ar]
            assert all(arena.data_ptr() <= p < arena.data_ptr() + 4 * arena.numel() for p in ptrs)       # zero-copy arena views
            assert ptrs == sorted(ptrs, reverse=True)                                                     # tail of the arena first
            assert sum(events[i][2] for i in ar) == arena.numel()
        else:
            assert all(events[i][3] == torch.bfloat16 for i in ar)
            assert sum(events[i][2] for i in ar) == ref.numel()
            assert torch.isfinite(ref).all() and float(ref.abs().sum()) > 0
