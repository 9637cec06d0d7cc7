"""Lightning-free data-parallel training loop for the DDPM pipeline: what scripts/training_ddp.py:185-206
of the reference intends (zero_grad; training_step; backward; optimizer.step), minus its bugs (SURVEY 2.4:
the reference bypasses DistributedDataParallel.forward, so its all-reduce never fires).

One process per GPU (torchrun env).  Gradients are averaged by ``parallel.GradBucketReducer`` while the
backward kernels are still running; the update is one fused HIP AdamW launch over the flat arena."""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from .optim import HipAdamW
from .parallel import GradBucketReducer, broadcast_parameters


class DPTrainer:
    def __init__(self, ddpm, lr: Optional[float] = None, bucket_bytes: int = 64 << 20, scale_lr_by_sqrt_world: bool = True,
                 comm_dtype: torch.dtype = torch.float32, device_timesteps: bool = True):
        self.ddpm = ddpm
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        broadcast_parameters(ddpm)
        engine = ddpm.backbone.engine()
        order = engine.param_order()
        kw = dict(ddpm.hparams.opt_kwargs)
        if lr is not None:
            kw["lr"] = lr
        kw.setdefault("lr", 1e-3)
        if scale_lr_by_sqrt_world:                       # abstract_diffusion.py:118
            kw["lr"] = kw["lr"] * (self.world ** 0.5)
        self.opt = HipAdamW(ddpm.parameters(), arena_order=order,
                            **{k: v for k, v in kw.items() if k in ("lr", "betas", "eps", "weight_decay")})
        self.opt.build_arena()                           # re-homes parameters and gradients into flat arenas
        self.reducer = GradBucketReducer(order, bucket_bytes=bucket_bytes, comm_dtype=comm_dtype)
        if device_timesteps and hasattr(ddpm, "device_timesteps"):
            ddpm.device_timesteps = True                 # t drawn by rho_randint: no CPU randint + H2D copy per step
        ddpm.backbone.grad_hooks = self.reducer
        ddpm.train()

    def step(self, batch) -> torch.Tensor:
        self.opt.zero_grad()
        loss = self.ddpm.training_step(batch)
        loss.backward()
        self.reducer.finish()
        self.opt.step()
        return loss
