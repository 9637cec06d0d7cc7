"""Layer primitives of the UNet (reference: rho_diffusion/layers.py:71-199), registered under the
same names.  The modules are parameter containers with the reference's ``state_dict`` layout;
inside ``UNet`` they are lowered to HIP kernel launches by ``engine.unet_engine``; called on their
own (NC* float32 GPU tensors) they run the same kernels through ``functional``."""
from __future__ import annotations

import torch
from torch import nn

from .registry import registry

__all__ = ["GroupNorm32", "conv_nd", "avg_pool_nd", "mean_flat", "zero_module", "normalization", "checkpoint"]


@registry.register_layer("GroupNorm32")
class GroupNorm32(nn.GroupNorm):
    """GroupNorm computed in fp32 and cast back (layers.py:71-74)."""

    def forward(self, x):
        from . import functional as HF
        return HF.group_norm32(x, self.num_groups, self.weight, self.bias, self.eps)


class _HipConvMixin:
    def forward(self, x):
        from . import functional as HF
        return HF.conv_nd(x, self.weight, self.bias, self.stride, self.padding)


class HipConv1d(_HipConvMixin, nn.Conv1d):
    pass


class HipConv2d(_HipConvMixin, nn.Conv2d):
    pass


class HipConv3d(_HipConvMixin, nn.Conv3d):
    pass


@registry.register_layer("conv_nd")
def conv_nd(dims, *args, **kwargs):
    """1-D / 2-D / 3-D convolution module (layers.py:77-88); ValueError for other dims."""
    if dims == 1:
        return HipConv1d(*args, **kwargs)
    elif dims == 2:
        return HipConv2d(*args, **kwargs)
    elif dims == 3:
        return HipConv3d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


class _HipAvgPoolMixin:
    """kernel = stride = 2 (3-D: (1, 2, 2)): the only pooling the UNet builds (unet_v2.py:153,165); runs rho_avgpool2x."""

    def forward(self, x):
        from . import functional as HF
        return HF.avg_pool(x, self.kernel_size, self.stride)


class HipAvgPool1d(_HipAvgPoolMixin, nn.AvgPool1d):
    pass


class HipAvgPool2d(_HipAvgPoolMixin, nn.AvgPool2d):
    pass


class HipAvgPool3d(_HipAvgPoolMixin, nn.AvgPool3d):
    pass


@registry.register_layer("avg_pool_nd")
def avg_pool_nd(dims, *args, **kwargs):
    """1-D / 2-D / 3-D average pooling module (layers.py:91-102); ValueError for other dims."""
    if dims == 1:
        return HipAvgPool1d(*args, **kwargs)
    elif dims == 2:
        return HipAvgPool2d(*args, **kwargs)
    elif dims == 3:
        return HipAvgPool3d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


@registry.register_layer("mean_flat")
def mean_flat(tensor):
    """layers.py:105-110: mean over all non-batch axes.  On the product path (GPU tensors) this is the HIP reduction
    rho_mean_flat (float32, fixed order); a host tensor is a host utility call and stays on torch (the engine never calls this)."""
    if tensor.is_cuda:
        from .engine import ops
        return ops.mean_flat(tensor).to(tensor.dtype)
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


def zero_module(module):
    """Zero all parameters of a module (layers.py:113-119)."""
    for p in module.parameters():
        p.detach().zero_()
    return module


def normalization(channels):
    """GroupNorm32 with 32 groups (layers.py:122-129)."""
    return GroupNorm32(32, channels)


def checkpoint(func, inputs, params, flag):
    """layers.py:153-168.  The HIP engine decides itself what to keep and what to recompute in
    backward (attention is always recomputed, as the reference does at unet_v2.py:334), so the flag
    only matters for code that calls this helper directly."""
    # (host-side control flow only: torch's checkpoint re-invokes ``func``, whatever ``func`` launches; no arithmetic of its own)
    if flag:
        from torch.utils.checkpoint import checkpoint as _ckpt
        return _ckpt(func, *inputs, use_reentrant=False)
    return func(*inputs)
