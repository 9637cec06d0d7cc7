"""Helpers called by the hot path's callers (reference: rho_diffusion/utils.py:45-81,166-220)."""
from __future__ import annotations

import hashlib
import itertools
import json
import os

import numpy as np
import torch


def ddp_setup(backend: str = "nccl"):
    """One process per GPU, ranks from the torchrun environment (reference: utils.py:45-81 reads
    Intel-MPI variables and initialises oneCCL; on ROCm ``"nccl"`` IS RCCL over xGMI).
    Returns (rank, local_rank, world_size)."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl" and torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def save_model_checkpoint(model, filename):
    """Backbone-only state_dict, reference key names (utils.py:166-167)."""
    torch.save(model.state_dict(), filename)


def save_samples_h5(filename, samples, name: str = "data") -> None:
    """scripts/inference.py:168-169: ``with h5py.File(fn, "w") as h5f: h5f["data"] = pred_images.cpu().numpy()`` - the file the
    reference's plotting reads back.  Written through ``h5io`` (system libhdf5 over ctypes, no h5py)."""
    from . import h5io
    arr = samples.detach().cpu().numpy() if torch.is_tensor(samples) else np.asarray(samples)
    h5io.write(filename, {name: arr}, mode="w")


def calculate_sha512_embedding(d: dict, l: int = 128):
    """utils.py:170-177: ASCII codes of the sha512 hex digest / 128, repeated to length l."""
    h = hashlib.sha512(json.dumps(d, sort_keys=True).encode()).hexdigest()
    return torch.tensor(np.array(h, "c").view(np.uint8).repeat(l // 128) / 128, dtype=torch.float32)


def sample_from_discrete_parameter_space(param_dict: dict, batch_size: int, random=True, device=None) -> torch.Tensor:
    """utils.py:213-220: rows of itertools.product over the parameter values."""
    keys, values = zip(*param_dict.items())
    combinations = torch.tensor([v for v in itertools.product(*values)], device=device)
    if random:
        idx = torch.randint(low=0, high=combinations.shape[0], size=(batch_size,), device=device)
    else:
        idx = torch.arange(start=0, end=batch_size, step=1, device=device)
    return combinations[idx]


def number_cast_dict(input_dict: dict) -> dict:
    """utils.py:223-244."""
    def cast(v):
        try:
            v = float(v)
            if v.is_integer():
                v = int(v)
        except (ValueError, TypeError):
            pass
        return v
    return {k: [cast(x) for x in v] if isinstance(v, list) else cast(v) for k, v in input_dict.items()}
