"""Exponential moving average of a model's weights, same class surface as the reference's
``rho_diffusion/ema.py:29-79`` (``ExponentialMovingAverage(model, decay=0.9999)``, ``update()``, ``forward`` and
``denoise_process`` on the shadow model, attributes ``ema_model`` / ``step_id`` / ``current_ema_frac`` / ``decay_func``).

MI355X form: the shadow weights live in ONE flat float32 arena laid out in the address order of the live parameters, so an
update is one ``rho_ema_update`` launch per contiguous run of live parameters - a single launch over all 166.8 M weights once
``HipAdamW`` has re-homed the model into its arena - instead of one launch per tensor.  Arithmetic per element (float32):
``shadow -= float32(1 - frac) * (shadow - param)`` with ``frac = decay * (1 - exp(-step / 2000))``, bit-identical to the
reference's tensor expression.  The shadow model is a copy WITHOUT the HIP engine state of ``UNet`` (plans, descriptors and raw
device pointers belong to the live model; the shadow builds its own engine on first use)."""
from __future__ import annotations

import copy
import math
from typing import List, Tuple

import numpy as np
import torch
from torch import nn

from . import hip

__all__ = ["ExponentialMovingAverage"]


class ExponentialMovingAverage(nn.Module):
    def __init__(self, model: nn.Module, decay: float = 0.9999) -> None:
        super().__init__()
        self.model = model
        self.decay = decay
        self.decay_func = self._fraction
        self.step_id = 0
        self.current_ema_frac = 0.0
        self.ema_model = copy.deepcopy(model).eval()          # UNet.__deepcopy__ leaves the engine state behind
        self._pairs: List[Tuple[nn.Parameter, nn.Parameter]] = []
        self._runs = None
        self._run_sig = None
        shadow = dict(self.ema_model.named_parameters())
        for name, p in model.named_parameters():
            if name not in shadow:
                raise KeyError(f"EMA: the copy has no parameter {name}")
            shadow[name].requires_grad_(False)
            self._pairs.append((p, shadow[name]))
        if len(shadow) != len(self._pairs):
            raise KeyError("EMA: parameter sets of the model and its copy differ")

    def _fraction(self, step: int) -> float:
        """Warm-up of the decay: decay * (1 - exp(-step / 2000))  (ema.py:36)."""
        return self.decay * (1 - math.exp(-step / 2000))

    # ------------------------------------------------------------------ flat shadow arena
    def _layout(self):
        """(Re)build the shadow arena in the address order of the live parameters and the list of contiguous runs
        [(shadow offset, live data_ptr, elements)].  Redone only when the live parameters moved (e.g. into an optimizer arena)."""
        sig = tuple(p.data_ptr() for p, _ in self._pairs)
        if self._runs is not None and sig == self._run_sig:
            return
        for p, sh in self._pairs:
            hip.require_gpu(p, "EMA parameter")
            if p.dtype != torch.float32 or sh.dtype != torch.float32 or not p.is_contiguous():
                raise hip.RhoHipError("EMA: parameters must be contiguous float32")
        order = sorted(range(len(self._pairs)), key=lambda i: self._pairs[i][0].data_ptr())
        total = sum(self._pairs[i][0].numel() for i in order)
        flat = torch.empty(total, dtype=torch.float32, device=self._pairs[0][0].device)
        runs = []
        off = 0
        for i in order:
            p, sh = self._pairs[i]
            k = p.numel()
            flat[off:off + k].copy_(sh.detach().reshape(-1))
            sh.data = flat[off:off + k].view_as(sh)
            if runs and runs[-1][1] + 4 * runs[-1][2] == p.data_ptr():
                runs[-1][2] += k                                   # adjacent in memory on both sides: extend the run
            else:
                runs.append([off, p.data_ptr(), k])
            off += k
        self._flat, self._runs, self._run_sig = flat, runs, sig

    @torch.no_grad()
    def update(self) -> None:
        self.step_id += 1
        frac = self.decay_func(self.step_id)
        self.current_ema_frac = frac
        self._layout()
        # (1.0 - frac) is a Python double; multiplying a float32 tensor by it rounds it to float32 first
        omf = float(np.float32(1.0 - frac))
        L = hip.lib()
        stream = hip.stream()
        base = self._flat.data_ptr()
        for off, src, k in self._runs:
            hip.check(L.rho_ema_update(base + 4 * off, src, k, omf, stream), "rho_ema_update")
        live = dict(self.model.named_buffers())
        for name, buf in self.ema_model.named_buffers():
            buf.copy_(live[name])

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)

    def denoise_process(self, *args, **kwargs):
        return self.ema_model.denoise_process(*args, **kwargs)
