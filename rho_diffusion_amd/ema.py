"""ExponentialMovingAverage of a model's parameters (reference: rho_diffusion/ema.py:29-79), same class surface:
``ExponentialMovingAverage(model, decay=0.9999)``, ``update()``, ``forward`` / ``denoise_process`` on the shadow model.
The update ``shadow -= (1 - frac) * (shadow - param)`` with ``frac = decay * (1 - exp(-step / 2000))`` runs in
librho_hip.so (rho_ema_update), one launch per parameter tensor (GPU tensors only); buffers are copied."""
from __future__ import annotations

import copy
import math
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from . import hip

__all__ = ["ExponentialMovingAverage"]


class ExponentialMovingAverage(nn.Module):
    def __init__(self, model: nn.Module, decay: float = 0.9999) -> None:
        super().__init__()
        self.model = model
        self.ema_model = copy.deepcopy(model).eval()
        self.decay_func = lambda x: decay * (1 - math.exp(-x / 2000))
        self.step_id = 0
        self.current_ema_frac = 0.0
        for param in self.ema_model.parameters():
            param.requires_grad_(False)

    @torch.no_grad()
    def update(self) -> None:
        self.step_id += 1
        current_frac = self.decay_func(self.step_id)
        self.current_ema_frac = current_frac
        model_params = OrderedDict(self.model.named_parameters())
        shadow_params = OrderedDict(self.ema_model.named_parameters())
        assert model_params.keys() == shadow_params.keys()
        # (1.0 - current_frac) is a Python double; multiplying a float32 tensor by it rounds it to float32 first
        omf = float(np.float32(1.0 - current_frac))
        L = hip.lib()
        stream = torch.cuda.current_stream().cuda_stream
        for name, param in model_params.items():
            sh = shadow_params[name]
            hip.require_gpu(param, name)
            if sh.dtype != torch.float32 or param.dtype != torch.float32 or not sh.is_contiguous() or not param.is_contiguous():
                raise hip.RhoHipError(f"EMA: parameter {name} must be contiguous float32")
            hip.check(L.rho_ema_update(sh.data_ptr(), param.data_ptr(), sh.numel(), omf, stream), "rho_ema_update")
        model_buffers = OrderedDict(self.model.named_buffers())
        shadow_buffers = OrderedDict(self.ema_model.named_buffers())
        assert model_buffers.keys() == shadow_buffers.keys()
        for name, buffer in model_buffers.items():
            shadow_buffers[name].copy_(buffer)

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)

    def denoise_process(self, *args, **kwargs):
        return self.ema_model.denoise_process(*args, **kwargs)
