"""Experiment configuration files (reference: rho_diffusion/config.py:36-110, the pydantic ``ExperimentConfig`` the scripts
load with ``ExperimentConfig.from_json``).

Plain ``json`` + explicit typing - no pydantic dependency - with the same attribute surface: ``config.model.name`` /
``config.model.kwargs`` (also ``dataset``, ``optimizer``, ``lr_scheduler``, ``noise_schedule``), ``config.training.*``,
``config.inference.*``.  Component kwargs go through ``number_cast_dict`` like the reference's validator (utils.py:223-244),
except that JSON booleans STAY booleans: under the pinned pydantic 1.10 they were coerced through ``str`` (``false`` -> the truthy
``"False"``), under pydantic 2 to 0 / 1; keeping ``false`` falsy is what the example files mean (``"use_new_attention_order":
false`` => QKVAttentionLegacy).  Unknown keys of the ``training`` / ``inference`` sections (``np``,
``benchmark_mode`` in the shipped examples) are dropped silently, as pydantic's default does.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, Union

from .utils import number_cast_dict

__all__ = ["ComponentConfig", "TrainingConfig", "InferenceConfig", "ExperimentConfig"]


class _Section:
    _fields: Dict[str, Any] = {}          # name -> default (``...`` = required)

    def __init__(self, **values):
        for key, default in self._fields.items():
            if key in values:
                setattr(self, key, self._convert(key, values[key]))
            elif default is ...:
                raise ValueError(f"{type(self).__name__}: field required: {key}")
            else:
                setattr(self, key, default)

    def _convert(self, key, value):
        return value

    def dict(self) -> dict:
        return {k: getattr(self, k) for k in self._fields}

    def __repr__(self) -> str:
        return f"{type(self).__name__}({', '.join(f'{k}={getattr(self, k)!r}' for k in self._fields)})"


def _cast_kwargs(kwargs: dict) -> dict:
    keep = {k: v for k, v in kwargs.items() if isinstance(v, bool) or (isinstance(v, list) and any(isinstance(x, bool) for x in v))}
    out = number_cast_dict({k: v for k, v in kwargs.items() if k not in keep})
    out.update(keep)
    return {k: out[k] for k in kwargs}          # original key order


class ComponentConfig(_Section):
    """A name + kwargs mapping (config.py:36-48)."""
    _fields = {"name": ..., "kwargs": ...}

    def _convert(self, key, value):
        if key == "name":
            if not isinstance(value, str):
                raise ValueError("ComponentConfig.name must be a string")
            return value
        if not isinstance(value, dict):
            raise ValueError("ComponentConfig.kwargs must be a mapping")
        return _cast_kwargs(value)


class TrainingConfig(_Section):
    """config.py:51-64."""
    _fields = {"device": ..., "loss_fn": "MSELoss", "ema_decay": 0.0, "batch_size": 32, "seed": 777, "min_epochs": 1,
               "max_epochs": 999, "save_checkpoint_every_n_epochs": 10, "sample_every_n_epochs": 5}

    def _convert(self, key, value):
        kind = type(self._fields[key]) if self._fields[key] is not ... else str
        return kind(value)


class InferenceConfig(_Section):
    """config.py:67-77."""
    _fields = {"device": ..., "checkpoint": ..., "parameter_space": ..., "cache_file": None, "plot_output_file": None, "seed": 777}

    def _convert(self, key, value):
        if key == "parameter_space" and not isinstance(value, dict):
            raise ValueError("InferenceConfig.parameter_space must be a mapping")
        return int(value) if key == "seed" else value


class ExperimentConfig(_Section):
    """config.py:80-110."""
    _fields = {"experiment": ..., "model": ..., "dataset": ..., "optimizer": ..., "lr_scheduler": ..., "noise_schedule": ...,
               "training": ..., "inference": ...}
    _kinds = {"model": ComponentConfig, "dataset": ComponentConfig, "optimizer": ComponentConfig, "lr_scheduler": ComponentConfig,
              "noise_schedule": ComponentConfig, "training": TrainingConfig, "inference": InferenceConfig}

    def _convert(self, key, value):
        kind = self._kinds.get(key)
        if kind is None:
            return str(value)
        if not isinstance(value, dict):
            raise ValueError(f"ExperimentConfig.{key} must be a mapping")
        return kind(**value)

    @classmethod
    def from_json(cls, json_path: Union[str, Path]) -> "ExperimentConfig":
        json_path = Path(json_path)
        if not json_path.exists():
            raise FileNotFoundError(f"Specified config file not found: {json_path}")
        with open(json_path) as read_file:
            return cls(**json.load(read_file))
