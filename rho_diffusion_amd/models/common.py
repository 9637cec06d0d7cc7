"""Timestep embedding layer of the registry (reference: rho_diffusion/models/common.py:27-80).

The sinusoid is evaluated by the HIP kernel ``rho_timestep_embed`` for any integer t (GPU tensors only: the product path has
no CPU fallback); the UNet engine calls the same kernel fused with the two ``time_embed`` linears."""
from __future__ import annotations

import torch
from torch import nn

from ..registry import registry


def sinosoidal_position_embedding(t: torch.Tensor, dim: int, wavelength: int = 10000) -> torch.Tensor:
    """float32 [len(t), dim] with columns (sin(t / w_0), cos(t / w_0), sin(t / w_1), ...), w_i = wavelength^(2i / dim)."""
    from .. import hip
    from ..engine import ops
    hip.require_gpu(t, "t")
    tt = t.reshape(-1).to(torch.int64).contiguous()
    return ops.timestep_embed(ops.sinusoid_frequencies(dim, wavelength, t.device), tt, tt.numel())


@registry.register_layer("SinusoidalPositionEmbedding")
class SinusoidalPositionEmbedding(nn.Module):
    def __init__(self, dim: int, wavelength: int = 10000) -> None:
        super().__init__()
        self.dim = dim
        self.wavelength = wavelength

    def forward(self, t: torch.Tensor):
        return sinosoidal_position_embedding(t, self.dim, self.wavelength)
