"""Synthetic training data of the reference: spherical-harmonic density fields
(rho_diffusion/data/synthetic.py:45-124 ``make_spherical_grid`` / ``compute_spherical_harmonic``, the random (l, m) draw of
``SphericalHarmonicDataset.random_set`` :240-253 and the per-sample layout of ``__getitem__`` :303).

The fields are generated ON THE DEVICE by ``rho_sph_harm_fields`` (csrc/sph_harm.hip: float64 evaluation of
|Y_l^{|m|}(theta, phi) * r| on the linspace(-2, 2, G)^3 grid, complex min-max normalisation, one cast to float32): no scipy, no
host arrays, no H2D copy - a training loop can draw fresh (l, m) every step.  GPU only, like the rest of the product path."""
from __future__ import annotations

import random
from typing import List, Sequence, Tuple

import torch

from . import hip
from .engine import ops
from .registry import registry
from .utils import calculate_sha512_embedding

__all__ = ["spherical_harmonic_fields", "spherical_harmonic_field", "SphericalHarmonicPool", "SphericalHarmonicDataset"]


def spherical_harmonic_fields(lm: Sequence[Tuple[int, int]], grid: int, dims: int = 3, device="cuda") -> torch.Tensor:
    """float32 ``[B, 1, G, G, G]`` (dims = 3) or the central z slice ``[B, 1, G, G]`` (dims = 2; the reference has no 2-D
    generator, SURVEY 8d states this choice) for the quantum numbers ``lm = [(l, m), ...]`` with |m| <= l."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise hip.RhoHipError("spherical_harmonic_fields runs on the GPU (rho_sph_harm_fields); there is no CPU path")
    for l, m in lm:
        if l < 0 or abs(m) > l:
            raise ValueError(f"invalid quantum numbers l={l}, m={m}")
    t = torch.tensor([[int(l), int(m)] for l, m in lm], dtype=torch.int32, device=dev)
    f = ops.sph_harm_fields(t, grid)                    # [B, G, G, G]: axes (y, x, z) as numpy.meshgrid(indexing="xy")
    if dims == 2:
        f = f[:, :, :, grid // 2].contiguous()
    elif dims != 3:
        raise ValueError("dims must be 2 or 3")
    return f[:, None]


def spherical_harmonic_field(l: int, m: int, grid: int, dims: int = 3, device="cuda") -> torch.Tensor:
    """One field, ``[1, G, G(, G)]`` (the layout of SphericalHarmonicDataset.__getitem__, synthetic.py:303)."""
    return spherical_harmonic_fields([(l, m)], grid, dims, device)[0]


class SphericalHarmonicPool:
    """A fixed pool of fields with random quantum numbers l ~ U{0..max_l}, m ~ U{-l..l} (synthetic.py:252-254),
    resident in HBM."""

    def __init__(self, grid: int, dims: int = 3, size: int = 8, max_l: int = 5, seed: int = 777, device="cuda"):
        rng = random.Random(seed)
        self.labels: List[Tuple[int, int]] = []
        for _ in range(size):
            l = rng.randint(0, max_l)
            self.labels.append((l, rng.randint(-l, l)))
        self.fields = spherical_harmonic_fields(self.labels, grid, dims, device)

    def batch(self, batch_size: int, device=None) -> torch.Tensor:
        idx = torch.arange(batch_size, device=self.fields.device) % len(self.fields)
        out = self.fields[idx].contiguous()
        return out.to(device) if device is not None else out


@registry.register_dataset("SphericalHarmonicDataset")
class SphericalHarmonicDataset(torch.utils.data.Dataset):
    """The reference's synthetic dataset (data/synthetic.py:127-305) with the fields generated on the device: item = (density
    float32 ``[1, G, G, G]`` on the GPU, label embedding = the 256-entry sha512 embedding of ``{"l": l, "m": m}``,
    synthetic.py:299-304).  (l, m) are drawn per item with ``random.randint`` exactly as ``random_set`` does (:240-254; the
    builtin RNG is seeded in the constructor like the reference's ``random_seed`` setter).  ``batch(B)`` draws B items with
    ONE generator launch - what a training loop on this engine should call.  HDF5 replay (``h5_path``) is not built.
    ``parameter_space`` is None, as in the reference (its constructor only binds a local)."""
    parameter_space = None

    def __init__(self, max_l, h5_path=None, length: int = 1000, random_seed=None, use_emb_as_labels: bool = True, device="cuda",
                 **grid_kwargs):
        if h5_path is not None:
            raise NotImplementedError("HDF5 replay is outside the hot path (SURVEY 8f row 3); fields are generated on the device")
        if isinstance(max_l, int):
            assert max_l > 0, f"Invalid maximum value of l > 0: {max_l}"
        self.max_l = max_l
        self.length = length
        self.use_emb_as_labels = use_emb_as_labels
        self.device = device
        self.grid_el = int(grid_kwargs.get("grid_el", 32))
        if not random_seed:
            import os
            random_seed = int(os.environ.get("PL_GLOBAL_SEED", 1616))
        self.random_seed = random_seed
        random.seed(random_seed)
        self.labels_emb_map = dict()

    @property
    def random_set(self) -> Tuple[int, int]:
        l = random.randint(0, self.max_l)
        return (l, random.randint(-l, l))

    def __len__(self) -> int:
        return self.length

    def _label(self, l: int, m: int) -> torch.Tensor:
        c = {"l": l, "m": m}
        emb = calculate_sha512_embedding(c, l=256)
        self.labels_emb_map[emb] = c
        return emb

    def __getitem__(self, index: int):
        l, m = self.random_set
        return spherical_harmonic_field(l, m, self.grid_el, 3, self.device), self._label(l, m)

    def batch(self, batch_size: int):
        lm = [self.random_set for _ in range(batch_size)]
        data = spherical_harmonic_fields(lm, self.grid_el, 3, self.device)
        labels = torch.stack([self._label(l, m) for l, m in lm]).to(data.device)
        return data, labels
