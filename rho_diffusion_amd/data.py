"""Synthetic training data of the reference: spherical-harmonic density fields
(rho_diffusion/data/synthetic.py:45-124 ``make_spherical_grid`` / ``compute_spherical_harmonic``, the random (l, m)
draw of ``SphericalHarmonicDataset.random_set`` :240-253 and the per-sample layout of ``__getitem__`` :303).

Host-side generator (numpy + scipy, exactly the reference's expressions): data generation is outside the GPU hot path;
the pool is built once and moved to the device, so neither the training loop nor the bench's timed region touches it.
"""
from __future__ import annotations

import random
from typing import List, Tuple

import numpy as np
import torch

__all__ = ["spherical_harmonic_field", "SphericalHarmonicPool"]


def _sph(m: int, l: int, theta: np.ndarray, phi: np.ndarray) -> np.ndarray:
    """scipy.special.sph_harm(m, n, theta, phi) of the reference (legacy argument order: azimuth first)."""
    try:
        from scipy.special import sph_harm_y      # scipy >= 1.15: (n, m, polar, azimuth)
        return sph_harm_y(l, m, phi, theta)
    except ImportError:  # pragma: no cover
        from scipy.special import sph_harm
        return sph_harm(m, l, theta, phi)


def spherical_harmonic_field(l: int, m: int, grid: int, dims: int = 3) -> torch.Tensor:
    """|Y_l^{|m|}(theta, phi) * r| on linspace(-2, 2, grid)^3, min-max normalised as a complex array
    (synthetic.py:115-124), float32 ``[1, G, G, G]``.  dims = 2 returns the central z slice ``[1, G, G]`` (the reference
    has no 2-D generator; SURVEY 8d states this choice)."""
    ax = np.linspace(-2, 2, grid)
    xg, yg, zg = np.meshgrid(ax, ax, ax, indexing="xy")
    with np.errstate(divide="ignore", invalid="ignore"):
        theta = np.arctan(np.sqrt(xg ** 2 + yg ** 2) / zg)
        phi = np.arctan(yg / xg)
    radial = np.sqrt(xg ** 2 + yg ** 2 + zg ** 2)
    sol = _sph(abs(m), l, theta, phi) * radial
    sol = (sol - sol.min()) / (sol.max() - sol.min())
    out = np.abs(sol).astype(np.float32)
    if dims == 2:
        out = out[:, :, grid // 2]
    return torch.from_numpy(out)[None]


class SphericalHarmonicPool:
    """A fixed pool of fields with random quantum numbers l ~ U{0..max_l}, m ~ U{-l..l} (synthetic.py:252-254)."""

    def __init__(self, grid: int, dims: int = 3, size: int = 8, max_l: int = 5, seed: int = 777):
        rng = random.Random(seed)
        self.labels: List[Tuple[int, int]] = []
        fields = []
        for _ in range(size):
            l = rng.randint(0, max_l)
            m = rng.randint(-l, l)
            self.labels.append((l, m))
            fields.append(spherical_harmonic_field(l, m, grid, dims))
        self.fields = torch.stack(fields)

    def batch(self, batch_size: int, device=None) -> torch.Tensor:
        idx = [i % len(self.fields) for i in range(batch_size)]
        out = self.fields[idx].contiguous()
        return out.to(device) if device is not None else out
