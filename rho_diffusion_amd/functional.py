"""Stand-alone use of the layer primitives on reference-layout tensors ([N, C, *spatial] float32 on
the GPU): each call packs to channels-last, runs the same HIP kernels the UNet engine uses, and
returns the reference layout.  Used when ``conv_nd`` / ``GroupNorm32`` / ``ResBlock`` /
``AttentionBlock`` / ``Upsample`` modules are called outside ``UNet`` (tests, custom backbones);
inside ``UNet`` the engine keeps activations channels-last end to end."""
from __future__ import annotations

from typing import Optional

import torch

from . import hip
from .engine import ops

Tensor = torch.Tensor
_DT = torch.float32  # stand-alone calls use the exact-f32 MFMA path


def _dims_of(x: Tensor) -> int:
    return x.dim() - 2


def _to_cl(x: Tensor, dtype=_DT) -> Tensor:
    hip.require_gpu(x, "input")
    return ops.pack_input(x.float().contiguous(), dtype)


def _k3(w: Tensor):
    k = list(w.shape[2:])
    while len(k) < 3:
        k.insert(0, 1)
    return tuple(int(v) for v in k)


def _stride_hw(stride, dims):
    s = list(stride) if isinstance(stride, (tuple, list)) else [stride] * dims
    while len(s) < 3:
        s.insert(0, 1)
    if s[0] != 1:
        raise hip.RhoHipError("depth stride must be 1 (unet_v2.py:153 uses (1,2,2))")
    return (int(s[1]), int(s[2]))


def _conv_cl(xcl, x2, weight, bias, *, stride_hw=(1, 1), up_hw=(0, 0), pre=None, pre_silu=False, res=None,
             channel_major_out=False):
    dt = xcl.dtype
    w = ops.prep_conv_weight(weight.detach().float().contiguous(), dt, cinp=xcl.shape[-1] + (x2.shape[-1] if x2 is not None else 0))
    cout = weight.shape[0]
    b = torch.zeros(w.shape[1], dtype=torch.float32, device=xcl.device)
    if bias is not None:
        b[:cout].copy_(bias.detach())
    split = 0 if channel_major_out else cout
    if not channel_major_out and w.shape[1] != cout:
        raise hip.RhoHipError("channels-last conv output needs cout % 32 == 0")
    return ops.conv(xcl, x2, w, b, kernel=_k3(weight), cout=cout, split=split, stride_hw=stride_hw, up_hw=up_hw,
                    pre_a=pre[0] if pre else None, pre_b=pre[1] if pre else None, pre_silu=pre_silu, res=res,
                    y2_dtype=torch.float32)


def conv_nd(x: Tensor, weight: Tensor, bias: Optional[Tensor], stride=1, padding=0) -> Tensor:
    """layers.conv_nd forward: kernel 1 (padding 0) or 3 (padding 1)."""
    dims = _dims_of(x)
    k = weight.shape[-1]
    pad = padding[0] if isinstance(padding, (tuple, list)) else padding
    if not ((k == 1 and pad == 0) or (k == 3 and pad == 1)):
        raise hip.RhoHipError("HIP conv supports kernel 1 / padding 0 and kernel 3 / padding 1")
    xcl = _to_cl(x)
    _, y2 = _conv_cl(xcl, None, weight, bias, stride_hw=_stride_hw(stride, dims), channel_major_out=True)
    N, Do, Ho, Wo = ops.conv_out_shape(xcl.shape, _k3(weight), _stride_hw(stride, dims), (0, 0))
    spatial = [Do, Ho, Wo][3 - dims:]
    return y2.view(N, weight.shape[0], *spatial)


def group_norm32(x: Tensor, groups: int, weight: Tensor, bias: Tensor, eps: float) -> Tensor:
    """layers.GroupNorm32 forward (32 groups, eps 1e-5): statistics kernel + identity 1x1 conv whose
    prologue applies the folded affine."""
    if groups != 32 or abs(eps - 1e-5) > 1e-12:
        raise hip.RhoHipError("HIP GroupNorm implements GroupNorm(32, C, eps=1e-5)")
    c = x.shape[1]
    xcl = _to_cl(x)
    a, b, _ = ops.gn_coeffs(xcl, None, weight.detach().float(), bias.detach().float())
    eye = torch.eye(c, device=x.device).view(c, c, *([1] * _dims_of(x)))
    _, y2 = _conv_cl(xcl, None, eye, None, pre=(a, b), channel_major_out=True)
    return y2.view_as(x).type(x.dtype)


def upsample_conv(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], dims: int) -> Tensor:
    """models.unet_v2.Upsample forward."""
    c = x.shape[1]
    if weight is None:
        weight = torch.zeros(c, c, *([3] * dims), device=x.device)
        weight[(torch.arange(c), torch.arange(c)) + tuple([1] * dims)] = 1.0   # identity 3x3: pure nearest upsample
    xcl = _to_cl(x)
    up = (1, 1) if dims >= 2 else (0, 1)
    _, y2 = _conv_cl(xcl, None, weight, bias, up_hw=up, channel_major_out=True)
    N, Do, Ho, Wo = ops.conv_out_shape(xcl.shape, _k3(weight), (1, 1), up)
    spatial = [Do, Ho, Wo][3 - dims:]
    return y2.view(N, weight.shape[0], *spatial)


def avg_pool(x: Tensor, kernel_size, stride) -> Tensor:
    """layers.avg_pool_nd forward for kernel = stride = 2 (3-D: (1, 2, 2))."""
    dims = _dims_of(x)
    ks = tuple(kernel_size) if isinstance(kernel_size, (tuple, list)) else (kernel_size,) * dims
    st = tuple(stride) if isinstance(stride, (tuple, list)) else (stride,) * dims
    want = (1, 2, 2) if dims == 3 else (2,) * dims
    if ks != want or st != want:
        raise hip.RhoHipError(f"HIP average pooling implements kernel = stride = {want} (what Downsample builds, unet_v2.py:153,165)")
    ycl = ops.avgpool2x(_to_cl(x), (1, 1) if dims >= 2 else (0, 1))
    return _from_cl(ycl, dims)[:, :x.shape[1]].contiguous()


def resblock(blk, x: Tensor, emb: Tensor) -> Tensor:
    """models.unet_v2.ResBlock forward (unet_v2.py:273-293), incl. the up / down form (:277-281)."""
    from torch import nn
    dims = blk.dims
    xcl = _to_cl(x)
    lin = blk.emb_layers[1]
    emb_out = ops.linear(emb.float().contiguous(), lin.weight.detach(), lin.bias.detach(), act_in=True)
    cout = blk.out_channels
    n0, n1 = blk.in_layers[0], blk.out_layers[0]
    a1, b1, _ = ops.gn_coeffs(xcl, None, n0.weight.detach(), n0.bias.detach())
    c_in, c_out = blk.in_layers[2], blk.out_layers[3]
    w1 = ops.prep_conv_weight(c_in.weight.detach(), _DT)
    radd = None if blk.use_scale_shift_norm else emb_out
    if getattr(blk, "updown", False):
        # h = in_conv(h_upd(SiLU(GN(x)))), x = x_upd(x): the activation is materialised, both tensors resampled as passes of their own
        from .models.unet_v2 import Upsample as _Up
        rs = (1, 1) if dims >= 2 else (0, 1)
        act = torch.empty_like(xcl)
        hip.check(hip.lib().rho_gn_apply(xcl.data_ptr(), xcl.shape[-1], None, 0, hip.dtype_code(xcl.dtype), xcl.shape[0],
                                         xcl.shape[1] * xcl.shape[2] * xcl.shape[3], a1.data_ptr(), b1.data_ptr(), 1, act.data_ptr(),
                                         hip.stream()), "rho_gn_apply")
        if isinstance(blk.h_upd, _Up):
            hh, xcl = ops.upsample2x(act, rs), ops.upsample2x(xcl, rs)
        else:
            hh, xcl = ops.avgpool2x(act, rs), ops.avgpool2x(xcl, rs)
        t1, _ = ops.conv(hh, None, w1, c_in.bias.detach(), kernel=_k3(c_in.weight), cout=cout, res_add=radd, res_add_stride=cout)
    else:
        t1, _ = ops.conv(xcl, None, w1, c_in.bias.detach(), kernel=_k3(c_in.weight), cout=cout, pre_a=a1, pre_b=b1,
                         pre_silu=True, res_add=radd, res_add_stride=cout)
    if blk.use_scale_shift_norm:
        a2, b2, _ = ops.gn_coeffs(t1, None, n1.weight.detach(), n1.bias.detach(), scale=emb_out, shift=emb_out[:, cout:],
                                  film_stride=2 * cout)
    else:
        a2, b2, _ = ops.gn_coeffs(t1, None, n1.weight.detach(), n1.bias.detach())
    if isinstance(blk.skip_connection, nn.Identity):
        sk = xcl
    else:
        sc = blk.skip_connection
        sk, _ = ops.conv(xcl, None, ops.prep_conv_weight(sc.weight.detach(), _DT), sc.bias.detach(), kernel=_k3(sc.weight), cout=cout)
    # final conv straight to the reference layout (channel-major f32), residual added channels-last first
    w2 = ops.prep_conv_weight(c_out.weight.detach(), _DT)
    out, _ = ops.conv(t1, None, w2, c_out.bias.detach(), kernel=_k3(c_out.weight), cout=cout, pre_a=a2, pre_b=b2,
                      pre_silu=True, res=sk)
    return _from_cl(out, dims)


def attention_block(blk, x: Tensor) -> Tensor:
    """models.unet_v2.AttentionBlock forward (unet_v2.py:336-342) -- bf16 attention kernel."""
    from .engine.unet_engine import UNetEngine
    dims = _dims_of(x)
    dt = torch.bfloat16
    xcl = _to_cl(x, dt)
    c = blk.channels
    a, b, _ = ops.gn_coeffs(xcl, None, blk.norm.weight.detach(), blk.norm.bias.detach())
    fake = type("E", (), {"device": x.device})()
    row_src = UNetEngine._qkv_row_src(fake, blk)
    wq = ops.prep_conv_weight(blk.qkv.weight.detach().contiguous(), dt, row_src=row_src)
    bq = blk.qkv.bias.detach()[row_src.long()].contiguous()
    N = x.shape[0]
    T = x.numel() // (N * c)
    qk, vt = ops.conv(xcl, None, wq, bq, kernel=(1, 1, 1), cout=3 * c, split=2 * c, pre_a=a, pre_b=b)
    ao = ops.attention(qk.view(N, T, 2 * c), vt, blk.num_heads)
    wp = ops.prep_conv_weight(blk.proj_out.weight.detach().contiguous(), dt)
    out, _ = ops.conv(ao.view(xcl.shape), None, wp, blk.proj_out.bias.detach(), kernel=(1, 1, 1), cout=c, res=xcl)
    return _from_cl(out, dims)


def _from_cl(ycl: Tensor, dims: int) -> Tensor:
    """channels-last [N, D, H, W, C] -> [N, C, *spatial] float32 via an identity 1x1 conv with
    channel-major output (keeps the data path on the HIP kernels)."""
    c = ycl.shape[-1]
    eye = ops.prep_conv_weight(torch.eye(c, device=ycl.device).view(c, c, 1).contiguous(), ycl.dtype)
    zb = torch.zeros(eye.shape[1], dtype=torch.float32, device=ycl.device)
    _, y2 = ops.conv(ycl, None, eye, zb, kernel=(1, 1, 1), cout=c, split=0, y2_dtype=torch.float32)
    N, D, H, W, _ = ycl.shape
    spatial = [D, H, W][3 - dims:]
    return y2.view(N, c, *spatial)
