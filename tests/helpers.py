"""Shared test helpers: golden loading, error metrics, template state_dicts."""
from __future__ import annotations

import os

import numpy as np
import torch

from golden_cfg import (ACT_CASES, DEEP_GALAXY_SPACE, PARAM_SPACE, UNET_CASES, UPDOWN_CASES, V1_CASES, WIDE_CASES,  # noqa: F401  (re-exported)
                        galaxy_labels)
from detdata import det_normal, det_state_dict, det_uniform  # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name))


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu().flatten()
    b = b.detach().double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def golden_template(g4, case: str) -> dict:
    """name -> empty tensor of the recorded shape (the reference's state_dict layout)."""
    tmpl = {}
    for item in g4[f"{case}/keys"]:
        k, shp = str(item).split("|")
        shape = tuple(int(s) for s in shp.split(",")) if shp else ()
        tmpl[k] = torch.empty(shape)
    return tmpl


def case_inputs(case: str):
    """(cfg, x, t, y) exactly as tests/golden/make_golden.py builds them."""
    kw, xshape, ykind = UNET_CASES[case]
    x = det_normal(xshape, case + "x")
    B = xshape[0]
    t = torch.tensor([(37 * i + 11) % 1000 for i in range(B)])
    y = None
    if ykind == "multi":
        keys = list(PARAM_SPACE.keys())
        y = torch.tensor([[PARAM_SPACE[k][(i + j) % len(PARAM_SPACE[k])] for j, k in enumerate(keys)] for i in range(B)],
                         dtype=torch.float32)
    elif ykind == "preemb":
        y = det_normal((B, 4 * kw["model_channels"]), case + "y")
    return dict(kw), x, t, y


def wide_case_inputs(case: str):
    """(cfg, x, t, y, parameter_space) of a g12 case (tests/golden/make_golden.py gen_g12)."""
    kw, xshape, ykind = WIDE_CASES[case]
    x = det_normal(xshape, case + "x")
    B = xshape[0]
    t = torch.tensor([(37 * i + 11) % 1000 for i in range(B)])
    y = torch.tensor(galaxy_labels(B), dtype=torch.float32) if ykind == "galaxy" else None
    return dict(kw), x, t, y, (DEEP_GALAXY_SPACE if ykind == "galaxy" else None)


def cosine(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu().flatten()
    b = b.detach().double().cpu().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def grad_digest_of(g: torch.Tensor) -> np.ndarray:
    g = g.detach().double().cpu().flatten()
    head = torch.zeros(6, dtype=torch.float64)
    head[: min(6, g.numel())] = g[:6]
    return np.concatenate([[g.norm().item(), g.sum().item()], head.numpy()])
