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

    Inside the UNet engine the lookup runs on the device (``rho_multi_embed``: category resolution + row sum in one launch,
    no host synchronisation; backward ``rho_multi_embed_bwd``).  ``forward`` below is the stand-alone module call for GPU
    label tensors and uses the same kernel; the module only holds the tables."""

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

    def _device_tables(self, dev):
        """Value lists of the parameter space (float32, concatenated), their offsets and the table pointers on the device; cached
        on the weights' storage (an optimizer arena or ``.to()`` re-homes them) like UNetEngine.cond_device_tables."""
        keys = list(self.embedding_layers.keys())
        weights = [self.embedding_layers[k].weight for k in keys]
        sig = (str(dev),) + tuple(w.data_ptr() for w in weights)
        cached = self.__dict__.get("_dev_tables")
        if cached is None or cached["sig"] != sig:
            vals = [torch.tensor(self.parameter_space[k]).to(torch.float32) for k in keys]    # the reference's conversion (:131)
            off = [0]
            for v in vals:
                off.append(off[-1] + v.numel())
            cached = dict(sig=sig, nkeys=len(keys), weights=weights, space=torch.cat(vals).to(dev),
                          key_off=torch.tensor(off, dtype=torch.int32, device=dev),
                          tables=torch.tensor([w.data_ptr() for w in weights], dtype=torch.int64, device=dev))
            self.__dict__["_dev_tables"] = cached
        return cached

    def forward(self, y: torch.Tensor) -> torch.Tensor:
        if len(self.embedding_layers) == 0:
            return None                                      # conditioning.py:117,139
        from .. import hip
        hip.require_gpu(y, "y")
        tabs = self._device_tables(y.device)
        return _MultiEmbedFunction.apply(y.to(torch.float32).contiguous(), self.embedding_dim, tabs, *tabs["weights"])


class _MultiEmbedFunction(torch.autograd.Function):
    """Stand-alone ``MultiEmbeddings.forward`` (callers outside the UNet engine: a user's own model, the sampling loop's
    pre-embedding): rho_multi_embed forward, rho_multi_embed_bwd into the table gradients, so the module trains under autograd
    like the reference's ``nn.Embedding`` sum.  An unknown label raises at once (conditioning.py:132)."""

    @staticmethod
    def forward(ctx, yf, dim, tabs, *weights):
        from .. import hip
        from ..hip import check, ptr
        B = yf.shape[0]
        out = torch.empty(B, dim, dtype=torch.float32, device=yf.device)
        idx = torch.empty(B, 16, dtype=torch.int32, device=yf.device)
        err = torch.zeros(1, dtype=torch.int32, device=yf.device)
        check(hip.lib().rho_multi_embed(ptr(yf), 1 if yf.dim() == 1 else yf.shape[1], ptr(tabs["space"]), ptr(tabs["key_off"]),
                                        ptr(tabs["tables"]), tabs["nkeys"], B, dim, ptr(out), ptr(idx), ptr(err), hip.stream()),
              "rho_multi_embed")
        if int(err.item()) & 2:
            raise IndexError("MultiEmbeddings: a label value is not in the parameter space")
        ctx.idx, ctx.nkeys, ctx.dim = idx, tabs["nkeys"], dim
        ctx.shapes = [tuple(w.shape) for w in weights]
        return out

    @staticmethod
    def backward(ctx, dout):
        from .. import hip
        from ..hip import check, ptr
        dout = dout.contiguous().float()
        grads = [torch.zeros(s, dtype=torch.float32, device=dout.device) for s in ctx.shapes]
        gp = torch.tensor([g.data_ptr() for g in grads], dtype=torch.int64, device=dout.device)
        check(hip.lib().rho_multi_embed_bwd(ptr(dout), ptr(ctx.idx), ptr(gp), ctx.nkeys, dout.shape[0], ctx.dim, hip.stream()),
              "rho_multi_embed_bwd")
        return (None, None, None, *grads)
