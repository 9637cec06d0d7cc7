"""Conditioning embedding (reference: rho_diffusion/models/conditioning.py:31-139)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Union

import torch
from torch import nn

from ..registry import registry


@registry.register_layer("MultiEmbeddings")
class MultiEmbeddings(nn.Module):
    """Sum over parameter-space keys of an ``nn.Embedding`` lookup, the category being the position
    of the label value in ``parameter_space[key]`` (exact float equality, conditioning.py:132).

    The lookup indices are resolved with a handful of tiny integer ops on the label tensor's
    device; the embedding rows are summed by the same ops.  ([B, k] labels, k <= a few keys:
    host-logic sized, not a kernel target - SURVEY 8a row a19.)"""

    def __init__(self, parameter_space=None, embedding_dim: int = 512, parameter_space_dim: int = 3,
                 embedding_size: Union[int, list, dict, OrderedDict] = None) -> None:
        super().__init__()
        self.embedding_layers = nn.ModuleDict()
        self.parameter_space = parameter_space
        self.embedding_dim = embedding_dim
        if parameter_space is not None and len(parameter_space) > 0:
            for key, value in self.parameter_space.items():
                self.embedding_layers[key] = nn.Embedding(num_embeddings=len(value), embedding_dim=self.embedding_dim)
        elif embedding_size is not None:
            if isinstance(embedding_size, int):
                for i in range(parameter_space_dim):
                    self.embedding_layers[str(i)] = nn.Embedding(embedding_size, embedding_dim)
            elif isinstance(embedding_size, list):
                for i in range(len(embedding_size)):
                    self.embedding_layers[str(i)] = nn.Embedding(embedding_size[i], embedding_dim)
            elif isinstance(embedding_size, dict):
                for key, value in embedding_size.items():
                    self.embedding_layers[key] = nn.Embedding(value, embedding_dim)

    def forward(self, y: torch.Tensor) -> torch.Tensor:
        emb = None
        for i, (key, layer) in enumerate(self.embedding_layers.items()):
            yi = y if y.dim() == 1 else y[:, i]
            space = torch.tensor(self.parameter_space[key], device=y.device)
            categorical = torch.where(yi[:, None] == space[None, :])[1]
            e = layer(categorical)
            emb = e if emb is None else emb + e
        return emb
