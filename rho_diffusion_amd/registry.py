"""Name -> class registry: the plugin seam of the reference (rho_diffusion/registry.py:27-203).

Same categories, decorators and ``registry.get(category, name)`` contract (KeyError for an
unknown name, AssertionError for an unknown category), pre-populated with the torch activations,
optimizers and ``nn`` modules the reference registers (:163-203), so pipelines and scripts that
resolve ``"UNetv2"``, ``"LinearSchedule"``, ``"MultiEmbeddings"``, ``"AdamW"``, ``"MSELoss"`` by
string work unchanged.
"""
from __future__ import annotations

from typing import Any, Callable, Dict

from torch import nn, optim

_CATEGORIES = ("models", "activations", "layers", "datasets", "nn", "schedules", "optimizers")


class Registry:
    mapping: Dict[str, Dict[str, Any]] = {c: {} for c in _CATEGORIES}

    @classmethod
    def _register(cls, category: str, name: str) -> Callable:
        def wrapper(target):
            cls.mapping[category][name] = target
            return target

        return wrapper

    @classmethod
    def register_model(cls, name: str) -> Callable:
        return cls._register("models", name)

    @classmethod
    def register_activation(cls, name: str) -> Callable:
        return cls._register("activations", name)

    @classmethod
    def register_layer(cls, name: str) -> Callable:
        return cls._register("layers", name)

    @classmethod
    def register_dataset(cls, name: str) -> Callable:
        return cls._register("datasets", name)

    @classmethod
    def register_nn(cls, name: str) -> Callable:
        return cls._register("nn", name)

    @classmethod
    def register_schedule(cls, name: str) -> Callable:
        return cls._register("schedules", name)

    @classmethod
    def register_optimizer(cls, name: str) -> Callable:
        return cls._register("optimizers", name)

    def get(self, category: str, name: str) -> Any:
        assert category in self.mapping, (
            f"{category} is not a category within Registry - valid entries: {self.mapping.keys()}.")
        target = self.mapping[category].get(name, None)
        if not target:
            raise KeyError(f"{name} is not a member of {category} category in Registry.")
        return target

    @property
    def categories(self):
        return list(self.mapping.keys())

    def __repr__(self) -> str:
        out = "Registry\n========"
        for key, value in self.mapping.items():
            out += f"{key}: {str(value)}\n"
        return out


registry = Registry()

for _name in ["ReLU", "SiLU", "Tanh", "Sigmoid", "ELU", "GELU", "PReLU", "Softmax", "LogSoftmax"]:
    registry.register_activation(_name)(getattr(nn, _name))

for _name in ["ASGD", "Adadelta", "Adagrad", "Adam", "AdamW", "Adamax", "LBFGS", "NAdam", "RAdam", "RMSprop",
              "Rprop", "SGD", "SparseAdam"]:
    registry.register_optimizer(_name)(getattr(optim, _name))

for _key in dir(nn):
    if _key != "Module":
        _cls = getattr(nn, _key, None)
        if isinstance(_cls, type) and issubclass(_cls, nn.Module):
            registry.register_nn(_key)(_cls)
