"""Legacy UNet ("UNet v1"; reference: rho_diffusion/models/unet.py:30-269), registered under the reference's names ``UNet``,
``UNetBlock2d``, ``UNetBlock3d`` with the same constructor arguments and ``state_dict`` layout (``time_mlp.1``, ``input_conv``,
``output_conv``, ``downsample.N.{time_embedding_readout, conv1, conv2, residual_conv, norm}``, ``upsample.N. ...``).

No script or shipped configuration of the reference instantiates it and its ``forward(data, t)`` lacks the ``y`` the
pipelines pass, so it is built as a drop-in model class for completeness, not as a tuned path: every operator is
one HIP launch (k_conv / k_wgrad for the convolutions and their gradients, csrc/unet_v1.hip for the block tail) wrapped in a
``torch.autograd.Function``; PyTorch carries the graph and owns the memory, no arithmetic runs in torch.

Structure (unet.py:117-135, 262-269): every block is
    h = act(conv1(x)); h = act(conv2(h)); h = h + residual_conv(x) + time_pe[:, :, None, None]; out = act(GroupNorm(8, C)(h))
at FULL resolution (all strides are 1: the "down" / "up" lists only change the channel count); "up" blocks take
``cat(x, skip)`` (read here as two source tensors, never concatenated) and use ConvTranspose for conv2 / residual_conv, which at
stride 1 / padding 1 is a plain convolution with flipped, transposed weights.  The first up block receives ``cat(x, x)``: the
reference pops the tensor it has just pushed (unet.py:264-267) - reproduced.

The 3-D block adds ``time_pe[..., None, None]`` = ``[B, C, 1, 1]`` to a ``[B, C, D, H, W]`` tensor (unet.py:128-129): that
broadcast pairs the batch axis with the channel axis and fails for every shape the model is meant for; ``UNetBlock3d`` is
registered and constructible like the reference's, and its forward raises the same kind of RuntimeError.
"""
from __future__ import annotations

from typing import List, Optional, Union

import torch
from torch import nn

from ..registry import registry
from .common import SinusoidalPositionEmbedding

__all__ = ["UNetBlock2d", "UNetBlock3d", "UNetV1", "UNet"]

_ACT_CODES = {nn.Identity: 0, nn.SiLU: 1, nn.ReLU: 2, nn.GELU: 3}


def _act_code(act: nn.Module) -> int:
    for cls, code in _ACT_CODES.items():
        if type(act) is cls:
            if cls is nn.GELU and getattr(act, "approximate", "none") != "none":
                break
            return code
    raise NotImplementedError(f"the HIP path of the legacy UNet implements Identity / SiLU / ReLU / GELU(erf), got {act!r}")


def _k3(w: torch.Tensor):
    k = [int(v) for v in w.shape[2:]]
    while len(k) < 3:
        k.insert(0, 1)
    return tuple(k)


def _cl_spatial(x_cl: torch.Tensor) -> int:
    return x_cl.shape[1] * x_cl.shape[2] * x_cl.shape[3]


# ----------------------------------------------------------------------------------------------------------- autograd ops
class _ConvFn(torch.autograd.Function):
    """y = conv(cat(x1, x2), weight) + bias (+ nc_add[n, c]), stride 1, padding k // 2; channels-last in and out, or (for the 1x1
    output convolution) float32 ``[N, cout, S]`` out.  ``weight`` is in torch's Conv layout [cout, cin, *k]."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, nc_add, channel_major_out):
        from ..engine import ops
        dt = x1.dtype
        c1 = x1.shape[-1]
        c2 = x2.shape[-1] if x2 is not None else 0
        w32 = weight.detach().float().contiguous()
        wp = ops.prep_conv_weight(w32, dt, cinp=c1 + c2)
        cout = weight.shape[0]
        b = torch.zeros(wp.shape[1], dtype=torch.float32, device=x1.device)
        b[:cout].copy_(bias.detach())
        kernel = _k3(weight)
        y, y2 = ops.conv(x1, x2, wp, b, kernel=kernel, cout=cout, split=0 if channel_major_out else cout,
                         res_add=nc_add, res_add_stride=nc_add.shape[1] if nc_add is not None else 0, y2_dtype=torch.float32)
        ctx.save_for_backward(x1, x2, w32, nc_add)
        ctx.meta = (kernel, cout, channel_major_out, wp, b)
        return y2 if channel_major_out else y

    @staticmethod
    def backward(ctx, dy):
        from ..engine import ops
        x1, x2, w32, nc_add = ctx.saved_tensors
        kernel, cout, channel_major, wp, b = ctx.meta
        dt = x1.dtype
        N, D, H, W, c1 = x1.shape
        c2 = x2.shape[-1] if x2 is not None else 0
        cin_real = w32.shape[1]
        wd = ops.prep_conv_weight_dgrad(w32, dt) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None
        ck = ops.elem_chunk(dt)
        dyw = ((cout + ck - 1) // ck) * ck
        if channel_major:          # [N, cout, S] float32 -> channels-last rows as wide as the dgrad weights expect
            dycl = ops.pack_input(dy.reshape(N, cout, D, H, W).float().contiguous(), dt, cpad=dyw)
        else:
            dycl = dy.contiguous()
        # weight + bias gradient (rho_conv_nd_wgrad: fp32 [taps, coutp, cinp] accumulation buffer, channel sums of dY)
        dfw = ops.make_conv_desc(x1, x2, wp, b, kernel=kernel, cout=cout, split=cout, y=dycl, y2=None)
        dw = torch.zeros(tuple(wp.shape), dtype=torch.float32, device=x1.device)
        db = torch.zeros(max(wp.shape[1], dycl.shape[-1]), dtype=torch.float32, device=x1.device)
        ops.conv_wgrad(dfw, dycl, dw, db)
        gw = torch.zeros(tuple(w32.shape), dtype=torch.float32, device=x1.device)
        ops.wgrad_finalize(dw, gw)
        gb = db[:cout].clone()
        gnc = None
        if nc_add is not None and ctx.needs_input_grad[4]:
            gnc = torch.empty(N, dycl.shape[-1], dtype=torch.float32, device=x1.device)
            ops.chan_sum(dycl, gnc)
            gnc = gnc[:, :cout].contiguous()
        g1 = g2 = None
        if wd is not None:
            if wd.shape[1] != c1 + c2 or wd.shape[2] != dycl.shape[-1] or cin_real != c1 + c2:
                raise RuntimeError("legacy UNet: data gradient needs channel counts that are multiples of 32")
            g1 = torch.empty_like(x1)
            g2 = torch.empty_like(x2) if x2 is not None else None
            zb = torch.zeros(wd.shape[1], dtype=torch.float32, device=x1.device)
            dd = ops.make_conv_desc(dycl, None, wd, zb, kernel=kernel, cout=c1 + c2, split=c1, y=g1, y2=g2, y2_cl=x2 is not None)
            ops.conv_launch(dd)
        return g1, g2, gw, gb, gnc, None


class _ActAddFn(torch.autograd.Function):
    """out = act(x) + r + nc[n, c]  (rho_act_add / rho_act_bwd)."""

    @staticmethod
    def forward(ctx, x, r, nc, act):
        from .. import hip
        from ..hip import check, ptr
        out = torch.empty_like(x)
        N, C = x.shape[0], x.shape[-1]
        check(hip.lib().rho_act_add(ptr(x), ptr(r), ptr(nc), ptr(out), hip.dtype_code(x.dtype), N, _cl_spatial(x), C, act, hip.stream()),
              "rho_act_add")
        ctx.save_for_backward(x)
        ctx.act = act
        ctx.has = (r is not None, nc is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .. import hip
        from ..engine import ops
        from ..hip import check, ptr
        (x,) = ctx.saved_tensors
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        check(hip.lib().rho_act_bwd(ptr(x), ptr(dout), ptr(dx), hip.dtype_code(x.dtype), x.numel(), ctx.act, hip.stream()), "rho_act_bwd")
        dnc = None
        if ctx.has[1]:
            dnc = torch.empty(x.shape[0], x.shape[-1], dtype=torch.float32, device=x.device)
            ops.chan_sum(dout, dnc)
        return dx, (dout if ctx.has[0] else None), dnc, None


class _GroupNormActFn(torch.autograd.Function):
    """y = act(GroupNorm(groups, C)(x) * gamma + beta)  (rho_groupnorm_act / rho_groupnorm_act_bwd)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, act):
        from .. import hip
        from ..hip import check, ptr
        N, C = x.shape[0], x.shape[-1]
        y = torch.empty_like(x)
        stats = torch.empty(N, groups, 2, dtype=torch.float32, device=x.device)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        check(hip.lib().rho_groupnorm_act(ptr(x), ptr(y), ptr(stats), ptr(g32), ptr(b32), hip.dtype_code(x.dtype), N, _cl_spatial(x), C,
                                          groups, float(eps), act, hip.stream()), "rho_groupnorm_act")
        ctx.save_for_backward(x, stats, g32, b32)
        ctx.meta = (groups, act)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import hip
        from ..hip import check, ptr
        x, stats, g32, b32 = ctx.saved_tensors
        groups, act = ctx.meta
        dy = dy.contiguous()
        N, C = x.shape[0], x.shape[-1]
        dx = torch.empty_like(x)
        dg = torch.zeros(C, dtype=torch.float32, device=x.device)
        db = torch.zeros(C, dtype=torch.float32, device=x.device)
        check(hip.lib().rho_groupnorm_act_bwd(ptr(x), ptr(dy), ptr(stats), ptr(g32), ptr(b32), ptr(dx), ptr(dg), ptr(db),
                                              hip.dtype_code(x.dtype), N, _cl_spatial(x), C, groups, act, hip.stream()),
              "rho_groupnorm_act_bwd")
        return dx, dg, db, None, None, None


class _LinearFn(torch.autograd.Function):
    """out = x @ w.T + b, float32  (rho_linear / rho_linear_bwd)."""

    @staticmethod
    def forward(ctx, x, w, b):
        from ..engine import ops
        x = x.float().contiguous()
        ctx.save_for_backward(x, w.detach().float().contiguous())
        return ops.linear(x, w.detach().float().contiguous(), b.detach().float().contiguous())

    @staticmethod
    def backward(ctx, dout):
        from ..engine import ops
        x, w = ctx.saved_tensors
        dout = dout.float().contiguous()
        dw, db = torch.zeros_like(w), torch.zeros(w.shape[0], dtype=torch.float32, device=w.device)
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        ops.linear_bwd(dout, x, w, dw, db, dx)
        return dx, dw, db


def _transposed_as_conv(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose weight [cin, cout, *k] at stride 1, padding (k - 1) / 2 -> the Conv weight [cout, cin, *k] of the same map:
    flip every spatial axis, swap the channel axes.  Pure data movement; autograd routes the gradient back the same way."""
    return w.flip(tuple(range(2, w.dim()))).transpose(0, 1).contiguous()


# ----------------------------------------------------------------------------------------------------------- modules
class AbstractUNetBlock(nn.Module):
    __conv_class__ = None
    __transpose_class__ = None
    __dims__ = 0

    def __init__(self, in_channels: int, out_channels: int, time_embedding_dim: int, is_up: bool = False, kernel_size: int = 3,
                 stride: int = 1, padding: int = 1, groups: int = 8, activation: Union[str, nn.Module] = "GELU",
                 residual: bool = False) -> None:
        super().__init__()
        self.time_embedding_readout = nn.Linear(time_embedding_dim, out_channels)
        conv, tconv = self.__conv_class__, self.__transpose_class__
        kw = dict(kernel_size=kernel_size, stride=stride, padding=padding)
        first_in = 2 * in_channels if is_up else in_channels
        self.conv1 = conv(in_channels=first_in, out_channels=out_channels, **kw)
        self.conv2 = (tconv if is_up else conv)(in_channels=out_channels, out_channels=out_channels, **kw)
        self.residual_conv = (tconv if is_up else conv)(in_channels=first_in, out_channels=out_channels, **kw) if residual else None
        self.norm = None if groups == 0 else nn.GroupNorm(groups, out_channels)
        if isinstance(activation, str):
            activation = registry.get("activations", activation)()
        self.activation = activation
        self.is_up = is_up
        if kernel_size not in (1, 3) or stride != 1 or padding != kernel_size // 2:
            raise NotImplementedError("the HIP path of the legacy UNet block implements kernel 3 / padding 1 (or 1 / 0) at stride 1 "
                                      "(unet.py builds nothing else)")

    def forward(self, x, t):
        raise RuntimeError("UNet v1 blocks run inside models.unet.UNetV1 (channels-last HIP path); call the model")

    def _conv_weight(self, mod: nn.Module) -> torch.Tensor:
        return _transposed_as_conv(mod.weight) if isinstance(mod, (nn.ConvTranspose2d, nn.ConvTranspose3d)) else mod.weight

    def run(self, x1, x2, time_pe):
        """x1 (, x2): channels-last activations; time_pe float32 [B, Tdim]; returns the block output, channels-last."""
        act = _act_code(self.activation)
        pe = _LinearFn.apply(time_pe, self.time_embedding_readout.weight, self.time_embedding_readout.bias)      # [B, cout]
        c1 = _ConvFn.apply(x1, x2, self._conv_weight(self.conv1), self.conv1.bias, None, False)
        a1 = _ActAddFn.apply(c1, None, None, act)
        c2 = _ConvFn.apply(a1, None, self._conv_weight(self.conv2), self.conv2.bias, None, False)
        if self.residual_conv is not None:
            r = _ConvFn.apply(x1, x2, self._conv_weight(self.residual_conv), self.residual_conv.bias, pe, False)    # + time_pe
            h = _ActAddFn.apply(c2, r, None, act)
        else:
            h = _ActAddFn.apply(c2, None, pe, act)
        if self.norm is None:
            return _ActAddFn.apply(h, None, None, act)
        return _GroupNormActFn.apply(h, self.norm.weight, self.norm.bias, self.norm.num_groups, self.norm.eps, act)


@registry.register_layer("UNetBlock2d")
class UNetBlock2d(AbstractUNetBlock):
    __conv_class__ = nn.Conv2d
    __transpose_class__ = nn.ConvTranspose2d
    __dims__ = 2


@registry.register_layer("UNetBlock3d")
class UNetBlock3d(AbstractUNetBlock):
    __conv_class__ = nn.Conv3d
    __transpose_class__ = nn.ConvTranspose3d
    __dims__ = 3


@registry.register_model("UNet")
class UNetV1(nn.Module):
    """unet.py:154-269.  Extra (optional) kwarg ``compute_dtype``: "fp32" (default: exact-f32 MFMA) or "bf16"."""

    def __init__(self, block_type: Union[str, type], input_channels: int, down_channels: List[int] = [64, 128, 256],
                 up_channels: List[int] = [256, 128, 64], time_embedding_dim: int = 32, kernel_size: int = 3, padding: int = 1,
                 activation: Union[str, nn.Module] = "ReLU", residual: bool = True, compute_dtype="fp32") -> None:
        super().__init__()
        if isinstance(block_type, str):
            block_type = registry.get("layers", block_type)
        self.time_mlp = nn.Sequential(SinusoidalPositionEmbedding(time_embedding_dim), nn.Linear(time_embedding_dim, time_embedding_dim))
        layer_type = nn.Conv3d if block_type == UNetBlock3d else nn.Conv2d
        self.input_conv = layer_type(in_channels=input_channels, out_channels=down_channels[0], kernel_size=3, stride=1, padding=1)
        self.output_conv = layer_type(up_channels[-1], input_channels, kernel_size=1, stride=1, padding=0)

        def blocks(chans, is_up):
            return nn.ModuleList([block_type(in_channels=chans[i], out_channels=chans[i + 1], time_embedding_dim=time_embedding_dim,
                                             is_up=is_up, kernel_size=kernel_size, padding=padding, activation=activation,
                                             residual=residual) for i in range(len(chans) - 1)])

        self.downsample = blocks(down_channels, False)
        self.upsample = blocks(up_channels, True)
        self.block_type = block_type
        self.input_channels = input_channels
        self.compute_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "f32": torch.float32}.get(compute_dtype, compute_dtype)

    @property
    def expected_dim(self) -> int:
        return 3 if self.block_type == UNetBlock2d else 4

    def forward(self, data: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        from .. import hip
        from ..engine import ops
        hip.require_gpu(data, "data")
        if self.block_type == UNetBlock3d:
            # unet.py:128-129: time_pe[..., None, None] is [B, C, 1, 1] against h [B, C, D, H, W]
            raise RuntimeError("UNetBlock3d: the reference adds time_pe of shape [B, C, 1, 1] to a [B, C, D, H, W] tensor "
                               "(unet.py:128-129); the shapes do not broadcast - the 3-D legacy UNet cannot run (use UNetv2)")
        if data.dim() != 4:
            raise RuntimeError(f"UNet (2-D blocks) expects [B, C, H, W], got {tuple(data.shape)}")
        dt = self.compute_dtype
        time_pe = _LinearFn.apply(self.time_mlp[0](t), self.time_mlp[1].weight, self.time_mlp[1].bias)
        x_cl = ops.pack_input(data.float().contiguous(), dt)                        # [B, 1, H, W, Cpad]; data carries no gradient
        x = _ConvFn.apply(x_cl, None, self.input_conv.weight, self.input_conv.bias, None, False)
        residual_h: List[torch.Tensor] = []
        for blk in self.downsample:
            x = blk.run(x, None, time_pe)
            residual_h.append(x)
        for blk in self.upsample:
            x = blk.run(x, residual_h.pop(), time_pe)                              # cat((x, skip), dim=1) as two sources
        y = _ConvFn.apply(x, None, self.output_conv.weight, self.output_conv.bias, None, True)     # float32 [B, Cout, S]
        return y.view(data.shape[0], self.input_channels, *data.shape[2:])


UNet = UNetV1        # ``from rho_diffusion.models.unet import UNet`` (the reference's module-level name) under install_alias()
