"""AdamW whose update runs in the fused HIP kernel (rho_adamw) over flat parameter / gradient /
moment arenas (reference: torch.optim.AdamW built at abstract_diffusion.py:103-119; defaults lr 1e-3,
betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2).  One launch per step instead of ~300 small ones;
7 x 4 B per parameter of HBM traffic."""
from __future__ import annotations

import torch

from .engine import ops


class HipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, arena_order=None):
        """``arena_order``: optional parameter order for the flat arena (``UNetEngine.param_order()``), so
        that the gradients that become final together during backward are adjacent in memory."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._arena = None
        self._arena_order = {id(p): i for i, p in enumerate(arena_order)} if arena_order is not None else None

    def _build_arena(self):
        """Re-home every parameter into one contiguous fp32 arena (views keep the module API intact)."""
        self._arena = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if self._arena_order is not None:
                ps.sort(key=lambda p: self._arena_order.get(id(p), 1 << 30))
            n = sum(p.numel() for p in ps)
            dev = ps[0].device
            flat = torch.empty(n, dtype=torch.float32, device=dev)
            grad = torch.zeros(n, dtype=torch.float32, device=dev)
            off = 0
            for p in ps:
                k = p.numel()
                flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + k].view_as(p)
                p.grad = grad[off:off + k].view_as(p)
                off += k
            self._arena.append(dict(flat=flat, grad=grad, m=torch.zeros_like(flat), v=torch.zeros_like(flat), step=0, params=ps))

    def build_arena(self):
        if self._arena is None:
            self._build_arena()
        return self._arena

    @property
    def flat_grads(self):
        if self._arena is None:
            self._build_arena()
        return [a["grad"] for a in self._arena]

    def zero_grad(self, set_to_none: bool = False):
        if self._arena is None:
            self._build_arena()
        for a in self._arena:
            a["grad"].zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self._arena is None:
            self._build_arena()
        for group, a in zip(self.param_groups, self._arena):
            # gradients produced outside the arena (first step / foreign autograd) are gathered once
            off = 0
            for p in a["params"]:
                k = p.numel()
                if p.grad is not None and p.grad.data_ptr() != a["grad"][off:off + k].data_ptr():
                    a["grad"][off:off + k].copy_(p.grad.reshape(-1))
                    p.grad = a["grad"][off:off + k].view_as(p)
                off += k
            a["step"] += 1
            b1, b2 = group["betas"]
            ops.adamw(a["flat"], a["grad"], a["m"], a["v"], group["lr"], b1, b2, group["eps"], group["weight_decay"], a["step"])
        # parameters changed in place through the arena: bump their version counters (no kernel) so
        # dependants (the engine's prepared conv weights) refresh
        for a in self._arena:
            for p in a["params"]:
                torch.autograd.graph.increment_version(p)
        return loss
