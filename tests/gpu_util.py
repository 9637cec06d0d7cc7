"""Helpers for the -m gpu parity tests (HIP path vs CPU oracle on the same seeded inputs)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

DEV = "cuda"


def to_cl(x: torch.Tensor, dtype) -> torch.Tensor:
    """[N, C, *S] -> channels-last [N, D, H, W, C] on the GPU (test-side plumbing)."""
    n, c = x.shape[:2]
    s = list(x.shape[2:])
    while len(s) < 3:
        s.insert(0, 1)
    y = x.reshape(n, c, *s).permute(0, 2, 3, 4, 1).contiguous()
    return y.to(DEV).to(dtype)


def from_cl(y: torch.Tensor, dims: int) -> torch.Tensor:
    """channels-last [N, D, H, W, C] -> [N, C, *S] float32 on the CPU."""
    n, d, h, w, c = y.shape
    out = y.float().cpu().permute(0, 4, 1, 2, 3)
    spatial = [d, h, w][3 - dims:]
    return out.reshape(n, c, *spatial).contiguous()


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).float()


def rnd(x: torch.Tensor, dtype) -> torch.Tensor:
    return bf16_round(x) if dtype == torch.bfloat16 else x


def tol(dtype, f32=2e-5, bf16=6e-3):
    return bf16 if dtype == torch.bfloat16 else f32
