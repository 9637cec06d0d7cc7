"""Noise schedules (reference: rho_diffusion/diffusion/schedule.py:34-221), same registry names,
constructor signatures and ``schedule[key]`` / ``schedule.dtype`` / ``schedule.to(device)`` protocol.

Tables are one-off host work (fp64 -> fp32, microseconds); the per-step gathers of the reference
(4 casts + 4 H2D copies per sampling step, abstract_diffusion.py:216-220) are replaced by one
device-resident coefficient table per schedule (``coef_table``) read inside the p_sample kernel.
"""
from __future__ import annotations

import math
from abc import ABC
from copy import deepcopy

import torch
from torch import Tensor
from torch.nn.functional import pad

from ..registry import registry

__all__ = ["LinearSchedule", "CosineBetaSchedule", "SigmoidSchedule"]

_KEYS = ("alpha_t", "beta_t", "alpha_bar_t", "sigma_t")


class AbstractSchedule(ABC):
    """schedule.py:34-138.  Values are stored fp32 and returned cast to ``self.dtype``."""

    @property
    def dtype(self) -> torch.dtype:
        return getattr(self, "_dtype", None) or torch.float32

    @dtype.setter
    def dtype(self, value) -> None:
        self._dtype = value or torch.float32

    @property
    def index(self) -> int:
        return self._index

    @index.setter
    def index(self, value: int) -> None:
        self._index = value

    def _get(self, name: str) -> Tensor:
        return getattr(self, "_" + name).type(self.dtype)

    alpha_t = property(lambda self: self._get("alpha_t"), lambda self, v: setattr(self, "_alpha_t", v))
    beta_t = property(lambda self: self._get("beta_t"), lambda self, v: setattr(self, "_beta_t", v))
    alpha_bar_t = property(lambda self: self._get("alpha_bar_t"), lambda self, v: setattr(self, "_alpha_bar_t", v))
    sigma_t = property(lambda self: self._get("sigma_t"), lambda self, v: setattr(self, "_sigma_t", v))

    @property
    def offset_alpha_bar_t(self) -> Tensor:
        return pad(self.alpha_bar_t[:-1], (1, 0), value=1.0)

    def state(self, index=None):
        if not index:
            index = self.index
        return {key: getattr(self, key)[index] for key in _KEYS}

    @property
    def last_state(self):
        return self._last_state

    @last_state.setter
    def last_state(self, state):
        self._last_state = deepcopy(state)

    def reset(self) -> None:
        self.index = 0
        self.last_state = {}

    def step(self) -> None:
        if not hasattr(self, "_index"):
            self.reset()
        else:
            self.last_state = self.state
            self.index += 1

    def __getitem__(self, key: str) -> Tensor:
        return getattr(self, key)

    def __len__(self) -> int:
        return len(self._alpha_bar_t)

    def __enter__(self):
        self.__old_dtype__ = self.dtype
        self.dtype = torch.float32

    def __exit__(self, *args, **kwargs):
        self.dtype = self.__old_dtype__

    def to(self, device) -> None:
        """The reference's ``to`` discards its results (schedule.py:133-138); here it warms the
        device-resident tables used by the HIP kernels."""
        if torch.device(device).type == "cuda":
            self.device_tables(device)

    # ---- device-side tables consumed by rho_q_sample / rho_p_sample_step
    def device_tables(self, device) -> dict:
        cache = self.__dict__.setdefault("_dev_tables", {})
        key = str(torch.device(device))
        if key not in cache:
            a, b, ab = self._alpha_t.float(), self._beta_t.float(), self._alpha_bar_t.float()
            # coefficients of ddpm.py:211-215, evaluated in fp32 exactly as the reference's tensor ops do
            coef = torch.stack([1 / a.sqrt(), b / (1 - ab).sqrt(), 0.8 * torch.sqrt(b)], dim=1).contiguous()
            cache[key] = {"alpha_bar": ab.contiguous().to(device), "coef": coef.to(device)}
        return cache[key]


def _sigma(alpha_bar64: Tensor, offset32: Tensor, beta64: Tensor) -> Tensor:
    return torch.sqrt((1 - offset32) / (1 - alpha_bar64) * beta64).to(torch.float32)


@registry.register_schedule("LinearSchedule")
class LinearSchedule(AbstractSchedule):
    """beta linear in [s*beta_1, s*beta_T], s = 1000/num_steps (schedule.py:141-168)."""

    def __init__(self, num_steps: int, beta_1: float = 1.0e-3, beta_T: float = 0.02, device="cpu") -> None:
        super().__init__()
        scale = 1000 / num_steps
        beta = torch.linspace(scale * beta_1, scale * beta_T, num_steps, dtype=torch.float64)
        alpha = 1.0 - beta
        alpha_bar = alpha.cumprod(0)
        self._beta_t = beta.to(torch.float32)
        self._alpha_t = alpha.to(torch.float32)
        self._alpha_bar_t = alpha_bar.to(torch.float32)
        offset = pad(self._alpha_bar_t[:-1], (1, 0), value=1.0)
        self._sigma_t = _sigma(alpha_bar, offset, beta)


@registry.register_schedule("CosineBetaSchedule")
class CosineBetaSchedule(AbstractSchedule):
    """Nichol & Dhariwal cosine schedule on num_steps + 1 points (schedule.py:171-214), including the
    reference's quirks: T+1 entries and sigma_t[0] = NaN."""

    def __init__(self, num_steps: int, offset: float = 0.008, device="cpu") -> None:
        super().__init__()
        t = torch.linspace(0.0, num_steps, num_steps + 1, dtype=torch.float64) / num_steps
        alpha_bar = torch.cos((t + offset) / (1 + offset) * math.pi * 0.5).pow(2.0)
        alpha_bar = alpha_bar.div(alpha_bar[0])
        ab32 = alpha_bar.to(torch.float32)
        ab32[ab32 < 0] = 0
        ab32[ab32 > 1] = 1
        self._alpha_bar_t = ab32
        off32 = pad(ab32[:-1], (1, 0), value=1.0)
        beta = (1 - (alpha_bar / off32)).clip_(0.0001, 0.9999)
        self._beta_t = beta.to(torch.float32)
        self._alpha_t = (1 - beta).to(torch.float32)
        self._sigma_t = _sigma(alpha_bar, off32, beta)


@registry.register_schedule("SigmoidSchedule")
class SigmoidSchedule(AbstractSchedule):
    def __init__(self, num_steps: int, offset: float = 0.008) -> None:
        super().__init__()
        raise NotImplementedError("SigmoidSchedule is not yet implemented.")
