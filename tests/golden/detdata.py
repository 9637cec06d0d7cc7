"""Exactly reproducible pseudo-random tensors (integer LCG -> float32), shared by the golden
generator and the tests so that multi-million-parameter weight sets never have to be stored:
only the small inputs/outputs live in ``tests/golden/*.npz``.  No torch/numpy RNG involved,
so the values are identical on every platform and library version."""
from __future__ import annotations

import zlib

import numpy as np
import torch


def det_uniform(shape, salt: str, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    seed = zlib.crc32(salt.encode()) & 0xFFFFFFFF
    i = np.arange(n, dtype=np.uint64)
    # two rounds of a 64-bit mix (splitmix64 finaliser), all arithmetic mod 2^64
    with np.errstate(over="ignore"):
        z = (i + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x632BE59BD9B4E019))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # 24-bit mantissa, exact in fp32
    out = (lo + (hi - lo) * u).astype(np.float32)
    return torch.from_numpy(out.reshape(shape))


def det_normal(shape, salt: str) -> torch.Tensor:
    """Approximately N(0,1) (sum of 4 uniforms, variance-matched); exact reproducibility is
    what matters, not the tail shape."""
    acc = sum(det_uniform(shape, f"{salt}#{k}", 0.0, 1.0) for k in range(4))
    return ((acc - 2.0) * float(np.sqrt(3.0))).to(torch.float32)


def det_state_dict(template: dict, salt: str) -> dict:
    """Fill every tensor of ``template`` (name -> tensor, only shapes are used) with
    deterministic values: norm weights ~1, biases small, everything else U(-b, b) with a
    fan-in bound so activations stay O(1).  Zero-initialised reference layers get real
    values too (a fresh UNetv2 outputs exactly 0 otherwise)."""
    out = {}
    for name, t in template.items():
        shape = tuple(t.shape)
        if name.endswith("weight") and len(shape) == 1:        # GroupNorm gamma
            out[name] = 1.0 + 0.2 * det_uniform(shape, salt + name)
        elif name.endswith("bias"):
            out[name] = 0.1 * det_uniform(shape, salt + name)
        elif "embedding_layers" in name:
            out[name] = 0.5 * det_uniform(shape, salt + name)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            out[name] = det_uniform(shape, salt + name) * float(np.sqrt(3.0 / fan_in))
    return out
