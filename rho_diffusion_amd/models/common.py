"""Timestep embedding (reference: rho_diffusion/models/common.py:27-80)."""
from __future__ import annotations

import torch
from torch import nn

from ..registry import registry


def sinosoidal_position_embedding(t: torch.Tensor, dim: int, wavelength: int = 10000) -> torch.Tensor:
    """Interleaved [sin(t/w_0), cos(t/w_0), sin(t/w_1), ...], always float32 (common.py:27-43).
    Host-side table builder: the UNet engine evaluates it once for t = 0..T-1 and gathers rows on
    the device (rho_embed_gather)."""
    assert dim % 2 == 0, "`dim` should be dividable by 2."
    device = t.device
    i = torch.arange(dim // 2, device=device)
    omega = torch.pow(wavelength, 2 * i / dim)
    pe = torch.empty(len(t), dim, device=device)
    pe[:, 2 * i] = torch.sin(t[:, None] / omega[None, :]).float()
    pe[:, 2 * i + 1] = torch.cos(t[:, None] / omega[None, :]).float()
    return pe


@registry.register_layer("SinusoidalPositionEmbedding")
class SinusoidalPositionEmbedding(nn.Module):
    def __init__(self, dim: int, wavelength: int = 10000) -> None:
        super().__init__()
        self.dim = dim
        self.wavelength = wavelength

    def forward(self, t: torch.Tensor):
        return sinosoidal_position_embedding(t, self.dim, self.wavelength)
