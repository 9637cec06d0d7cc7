"""torch.autograd glue: the UNet forward/backward run in the HIP engine, autograd only carries the
gradient of the prediction in and out.  Parameter gradients are accumulated by the kernels directly
into ``p.grad`` (so ``loss.backward(); optimizer.step()`` behaves as with stock autograd), which is why
the Function returns None for them."""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
from torch import nn

from .engine import ops


class UNetFunction(torch.autograd.Function):
    """pred = UNet(x, t, y).  ``anchor`` is any parameter that requires grad: it only makes autograd call
    ``backward``; ``hooks`` is an optional object with ``on_ready(params)`` for gradient all-reduce overlap."""

    @staticmethod
    def forward(ctx, x, anchor, engine, timesteps, y, hooks):
        ctx.engine = engine
        ctx.hooks = hooks
        out = engine.forward(x, timesteps, y, train=True)
        return out.clone()

    @staticmethod
    def backward(ctx, dout):
        cb = ctx.hooks.on_ready if ctx.hooks is not None else None
        ctx.engine.backward(dout.contiguous().float(), on_ready=cb)
        return None, None, None, None, None, None


class MSELossFunction(torch.autograd.Function):
    """mean((pred - target)^2) with the HIP reduction kernel (nn.MSELoss at ddpm.py:280)."""

    @staticmethod
    def forward(ctx, pred, target):
        loss, grad = ops.mse(pred.contiguous().float(), target.contiguous().float(), want_grad=True)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        # g is the scalar upstream gradient (1.0 for loss.backward()); applied on the device, no host sync
        from . import hip
        gs = g.reshape(1).float().contiguous()
        hip.check(hip.lib().rho_scale_by_device_scalar(grad.data_ptr(), gs.data_ptr(), grad.numel(), hip.stream()),
                  "rho_scale_by_device_scalar")
        return grad, None


def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return MSELossFunction.apply(pred, target)
