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
    """The reference's synthetic dataset (data/synthetic.py:127-348) with the fields generated on the device: item = (density
    float32 ``[1, G, G, G]`` on the GPU, label embedding = the 256-entry sha512 embedding of ``{"l": l, "m": m}``,
    synthetic.py:299-304).  (l, m) are drawn per item with ``random.randint`` exactly as ``random_set`` does (:240-254; the
    builtin RNG is seeded in the constructor like the reference's ``random_seed`` setter).  ``batch(B)`` draws B items with
    ONE generator launch - what a training loop on this engine should call.

    HDF5 (round 3): ``h5_path`` replays pre-computed samples from disk instead of generating them (:258-261, :285-289: datasets
    ``density``, ``l``, ``m``), ``to_hdf5`` writes such a file (:307-333) and ``from_hdf5`` opens one (:335-348) - through
    ``h5io`` (the system libhdf5 bound with ctypes; h5py is not needed).  Rows are read one hyperslab at a time.
    ``parameter_space`` is None, as in the reference (its constructor only binds a local)."""
    parameter_space = None

    def __init__(self, max_l, h5_path=None, length: int = 1000, random_seed=None, use_emb_as_labels: bool = True, device="cuda",
                 **grid_kwargs):
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
        self.h5_path = h5_path
        self._cursor = 0

    @property
    def h5_path(self):
        return self._h5_path

    @h5_path.setter
    def h5_path(self, value) -> None:
        import os
        from pathlib import Path
        if isinstance(value, str):
            value = Path(value)
        if isinstance(value, Path):
            assert value.exists(), f"{value} passed as target HDF5 file but was not found."      # synthetic.py:181-184
        self._h5_path = value

    @property
    def random_set(self) -> Tuple[int, int]:
        l = random.randint(0, self.max_l)
        return (l, random.randint(-l, l))

    def __len__(self) -> int:
        if self.h5_path:
            from . import h5io
            return h5io.shape(self.h5_path, "density")[0]
        return self.length

    def _label(self, l: int, m: int) -> torch.Tensor:
        c = {"l": int(l), "m": int(m)}
        emb = calculate_sha512_embedding(c, l=256)
        self.labels_emb_map[emb] = c
        return emb

    def _replay(self, index) -> Tuple[torch.Tensor, list]:
        """Rows ``index`` (int or unit-stride slice) of the file: densities [n, 1, G, G, G] on the device + their (l, m)."""
        from . import h5io
        import numpy as np
        d = h5io.read(self.h5_path, "density", index)
        l = np.atleast_1d(h5io.read(self.h5_path, "l", index))
        m = np.atleast_1d(h5io.read(self.h5_path, "m", index))
        t = torch.from_numpy(np.ascontiguousarray(d, dtype=np.float32))
        if isinstance(index, int):
            t = t[None]
        if t.dim() == 4:                       # [n, G, G, G] -> channel axis (files written by the reference carry it already)
            t = t[:, None]
        return t.to(self.device), [(int(a), int(b)) for a, b in zip(l, m)]

    def __getitem__(self, index: int):
        if self.h5_path:
            t, lm = self._replay(int(index))
            return t[0], self._label(*lm[0])
        l, m = self.random_set
        return spherical_harmonic_field(l, m, self.grid_el, 3, self.device), self._label(l, m)

    def batch(self, batch_size: int):
        if self.h5_path:
            n = len(self)
            lo = self._cursor if self._cursor + batch_size <= n else 0
            self._cursor = lo + batch_size
            data, lm = self._replay(slice(lo, lo + batch_size))
        else:
            lm = [self.random_set for _ in range(batch_size)]
            data = spherical_harmonic_fields(lm, self.grid_el, 3, self.device)
        labels = torch.stack([self._label(l, m) for l, m in lm]).to(data.device)
        return data, labels

    def to_hdf5(self, h5_path, chunk: int = 64) -> None:
        """synthetic.py:307-333: ``len(self)`` samples -> datasets ``density`` float32 [N, G, G, G], ``l``, ``m`` and the attribute
        ``seed``; '.h5' is appended when missing and an existing file is an error (h5py mode "x").  (The reference's own writer
        indexes its item tuples by string and cannot run; this is the file its reader :285-289 expects.)"""
        from pathlib import Path
        import numpy as np
        from . import h5io
        path = Path(h5_path).with_suffix(".h5")
        n = self.length if not self.h5_path else len(self)
        dens, ls, ms = [], [], []
        for lo in range(0, n, chunk):
            b = min(chunk, n - lo)
            if self.h5_path:
                data, lm = self._replay(slice(lo, lo + b))
            else:
                lm = [self.random_set for _ in range(b)]
                data = spherical_harmonic_fields(lm, self.grid_el, 3, self.device)
            dens.append(data[:, 0].cpu().numpy())
            ls += [a for a, _ in lm]
            ms += [c for _, c in lm]
        h5io.write(path, {"density": np.concatenate(dens, 0), "l": np.asarray(ls, dtype=np.int64), "m": np.asarray(ms, dtype=np.int64)},
                   attrs={"seed": int(self.random_seed)}, mode="x")

    @classmethod
    def from_hdf5(cls, h5_path, **kwargs) -> "SphericalHarmonicDataset":
        return cls(max_l=None, h5_path=h5_path, **kwargs)
