"""Data-parallel gradient averaging, one process per GPU (reference: DDP wrap at xpu.py:395-413 /
training_ddp.py:171-173 = PyTorch's bucketed all-reduce over oneCCL).  Here: RCCL over xGMI through
``torch.distributed`` (backend "nccl" IS RCCL on ROCm; "gloo" for the CPU tests).

The engine finalises parameter gradients from the tail of ``engine.param_order()`` to its head and
reports them through ``on_ready``.  Buckets are contiguous ranges of that order (contiguous memory when
the gradients live in the optimizer's flat arena), closed from the tail; a bucket's all-reduce is issued
the moment its last gradient is final, so the collective runs on RCCL's stream underneath the remaining
backward kernels.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): large buckets (default 64 MiB)
keep the ring near link bandwidth; the whole 3-D mc=64 model is 667 MB of fp32 gradients = ~11 buckets.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import nn


class GradBucketReducer:
    def __init__(self, params_in_order: Sequence[nn.Parameter], bucket_bytes: int = 64 << 20, group=None,
                 comm_dtype: torch.dtype = torch.float32):
        """``comm_dtype=torch.bfloat16`` sends the buckets as bf16 (334 instead of 667 MB per step at 3-D mc = 64):
        one cast pass per bucket on each side of the collective, the mean is rounded to bf16 once; fp32 gradients stay the
        optimizer's input.  Default float32 = the reference's DDP numerics."""
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("comm_dtype must be float32 or bfloat16")
        self.comm_dtype = comm_dtype
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # RCCL averages in the collective itself; gloo (CPU tests) has no AVG: sum, then scale
        self.avg_in_collective = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.params = list(params_in_order)
        self.index = {id(p): i for i, p in enumerate(self.params)}
        # buckets over the order, closed from the tail (that is where backward starts)
        self.buckets: List[List[int]] = []
        cur: List[int] = []
        size = 0
        for i in range(len(self.params) - 1, -1, -1):
            cur.append(i)
            size += self.params[i].numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {}
        for b, idxs in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        # bucket-close trace (bench.py, N = 1 "ddp_overlap"): when a list, every bucket close appends (bucket, bytes, HIP event
        # recorded on the launch stream at the point where the bucket's all-reduce would be issued) - also with world == 1,
        # where nothing is sent.  The measured position of each close inside the backward is the overlap budget of DESIGN section 6.
        self.trace: Optional[list] = None
        self.reset()

    def reset(self) -> None:
        self.pending = [len(b) for b in self.buckets]
        self.ready = [False] * len(self.params)
        self.works = []
        self.launched = [False] * len(self.buckets)

    # engine callback ------------------------------------------------------------------
    def on_ready(self, params: Sequence[nn.Parameter]) -> None:
        if self.world == 1 and self.trace is None:
            return
        for p in params:
            i = self.index.get(id(p))
            if i is None or self.ready[i]:
                continue
            self.ready[i] = True
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                if self.trace is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    self.trace.append((b, sum(self.params[j].numel() * 4 for j in self.buckets[b]), ev))
                if self.world > 1:
                    self._launch(b)

    def _flat_view(self, idxs: List[int]) -> Optional[torch.Tensor]:
        """One tensor covering the bucket if its gradients are adjacent in memory (optimizer arena)."""
        gs = [self.params[i].grad for i in sorted(idxs)]
        base = gs[0]
        end = base.data_ptr() + base.numel() * 4
        for g in gs[1:]:
            if g.data_ptr() != end or not g.is_contiguous():
                return None
            end += g.numel() * 4
        total = sum(g.numel() for g in gs)
        try:
            return torch.as_strided(base.reshape(-1), (total,), (1,))
        except RuntimeError:
            return None

    def _launch(self, b: int) -> None:
        idxs = self.buckets[b]
        self.launched[b] = True
        flat = self._flat_view(idxs)
        if flat is not None and self.comm_dtype == torch.float32:
            self.works.append((dist.all_reduce(flat, op=self._op(), group=self.group, async_op=True), flat, None))
        elif flat is not None:
            packed = flat.to(self.comm_dtype)                    # one cast pass; the fp32 arena view receives the mean in finish()
            self.works.append((dist.all_reduce(packed, op=self._op(), group=self.group, async_op=True), packed, [flat]))
        else:
            gs = [self.params[i].grad for i in idxs]
            packed = torch.cat([g.reshape(-1) for g in gs]).to(self.comm_dtype)      # copy-in / copy-out fallback (no arena)
            self.works.append((dist.all_reduce(packed, op=self._op(), group=self.group, async_op=True), packed, gs))

    def _op(self):
        return dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM

    def finish(self) -> None:
        """Wait for every bucket (launching any that never filled, e.g. parameters without gradients),
        turn sums into means, and re-arm for the next step."""
        if self.world > 1:
            for b in range(len(self.buckets)):
                if not self.launched[b]:
                    for i in self.buckets[b]:
                        if self.params[i].grad is None:
                            self.params[i].grad = torch.zeros_like(self.params[i])
                    self._launch(b)
            inv = 1.0 / self.world
            for work, flat, gs in self.works:
                work.wait()
                if not self.avg_in_collective:
                    flat.mul_(inv)
                if gs is not None:
                    off = 0
                    for g in gs:
                        g.copy_(flat[off:off + g.numel()].view_as(g))
                        off += g.numel()
        self.reset()


def broadcast_parameters(module: nn.Module, src: int = 0, group=None) -> None:
    """Rank-0 parameters to every rank (what the DDP constructor does, xpu.py:411)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in module.parameters():
            dist.broadcast(p.data, src=src, group=group)
        for b in module.buffers():
            dist.broadcast(b.data, src=src, group=group)
