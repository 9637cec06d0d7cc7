"""n-D UNet backbone, registered as ``"UNetv2"`` (reference: rho_diffusion/models/unet_v2.py:439-732).

Same constructor kwargs, same ``forward(x, timesteps, y=None)`` protocol, same ``state_dict`` keys
and shapes so reference checkpoints load unchanged.  The module tree only *holds
parameters*: ``UNet.forward`` hands the whole network to ``engine.unet_engine.UNetEngine``, which
lowers it to hand-written HIP kernels (conv / GroupNorm / attention / embedding) with channels-last
activations that never round-trip through PyTorch ops.
"""
from __future__ import annotations

import os

from abc import abstractmethod
from typing import Optional

import torch
import torch.nn as nn

from ..layers import avg_pool_nd, conv_nd, normalization, zero_module
from ..registry import registry


class TimestepBlock(nn.Module):
    """Any module whose forward takes the timestep embedding as 2nd argument (unet_v2.py:76-85)."""

    @abstractmethod
    def forward(self, x, emb):
        ...


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """Routes ``emb`` to the children that accept it (unet_v2.py:88-100)."""

    def forward(self, x, emb):
        for layer in self:
            x = layer(x, emb) if isinstance(layer, TimestepBlock) else layer(x)
        return x


class Upsample(nn.Module):
    """Nearest x2 (3-D: H and W only) + optional 3x3 conv (unet_v2.py:103-134)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if use_conv:
            self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=1)

    def forward(self, x):
        from .. import functional as HF
        assert x.shape[1] == self.channels
        return HF.upsample_conv(x, self.conv.weight if self.use_conv else None,
                                self.conv.bias if self.use_conv else None, self.dims)


class Downsample(nn.Module):
    """Stride-2 3x3 conv (3-D: stride (1,2,2)) or average pool (unet_v2.py:137-169)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        stride = 2 if dims != 3 else (1, 2, 2)
        if use_conv:
            self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=stride, padding=1)
        else:
            assert self.channels == self.out_channels
            self.op = avg_pool_nd(dims, kernel_size=stride, stride=stride)

    def forward(self, x):
        assert x.shape[1] == self.channels
        return self.op(x)


def activation_code(activation: nn.Module) -> int:
    """Activation code of the HIP engine (``engine.ops.ACT_CODES``) for an activation module of the reference's registry
    (rho_diffusion/registry.py:162-170, resolved at unet_v2.py:518-519).  The elementwise, parameter-free ones are built: SiLU (fused
    into the conv loaders), ReLU, GELU (erf form), Tanh, Sigmoid, ELU (alpha = 1) - the latter five run in the materialising GroupNorm
    passes.  PReLU carries a learnable parameter shared by every use of the one module instance, Softmax / LogSoftmax are not
    elementwise: refused, naming the line."""
    from ..engine.ops import ACT_CODES
    name = type(activation).__name__
    ok = name in ACT_CODES and name != "Identity"
    if name == "GELU" and getattr(activation, "approximate", "none") != "none":
        ok = False
    if name == "ELU" and float(getattr(activation, "alpha", 1.0)) != 1.0:
        ok = False
    if not ok:
        raise NotImplementedError(
            f"activation={name}: the reference resolves it through the registry (rho_diffusion/models/unet_v2.py:518-519, "
            "registry.py:162-170); the HIP engine builds SiLU, ReLU, GELU (erf), Tanh, Sigmoid and ELU (alpha = 1) for UNetv2 - not "
            "PReLU (a learnable parameter shared by all uses of the module), Softmax / LogSoftmax (not elementwise) or variants")
    return ACT_CODES[name]


class ResBlock(TimestepBlock):
    """GN-SiLU-conv, FiLM (scale-shift) or additive embedding, GN-SiLU-conv(zero-init), skip
    (unet_v2.py:172-293).  ``up`` / ``down`` (resblock_updown): the activated input and the skip input are resampled
    (nearest x2 / average pool) before the first conv (:277-281)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False,
                 activation=nn.SiLU()):
        super().__init__()
        if not (0.0 <= float(dropout) < 1.0):
            raise ValueError(f"dropout probability has to be in [0, 1), got {dropout} (nn.Dropout at unet_v2.py:239)")
        self.act_code = activation_code(activation)
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        self.dims = dims

        self.in_layers = nn.Sequential(normalization(channels), activation,
                                       conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.updown = up or down
        if up:
            self.h_upd = Upsample(channels, False, dims)
            self.x_upd = Upsample(channels, False, dims)
        elif down:
            self.h_upd = Downsample(channels, False, dims)
            self.x_upd = Downsample(channels, False, dims)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(
            activation,
            nn.Linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(
            normalization(self.out_channels), activation, nn.Dropout(p=dropout),
            zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def forward(self, x, emb):
        from .. import functional as HF
        return HF.resblock(self, x, emb)


class AttentionBlock(nn.Module):
    """Spatial self-attention (unet_v2.py:296-342); ``use_new_attention_order`` selects the
    channel->head mapping of QKVAttention (:400-436) instead of QKVAttentionLegacy (:365-397)."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False,
                 use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0, (
                f"q,k,v channels {channels} is not divisible by num_head_channels {num_head_channels}")
            self.num_heads = channels // num_head_channels
        self.use_checkpoint = use_checkpoint
        self.use_new_attention_order = bool(use_new_attention_order)
        self.norm = normalization(channels)
        self.qkv = conv_nd(1, channels, channels * 3, 1)
        self.proj_out = zero_module(conv_nd(1, channels, channels, 1))

    def forward(self, x):
        from .. import functional as HF
        return HF.attention_block(self, x)


@registry.register_model("UNetv2")
class UNet(nn.Module):
    """unet_v2.py:439-732.  Extra (optional) kwarg ``compute_dtype``: "bf16" (default; bf16
    storage + fp32 accumulation on MFMA) or "fp32" (exact-f32 MFMA path used for parity)."""

    def __init__(self, data_shape, in_channels: int, model_channels: int, out_channels: int, num_res_blocks: int,
                 attention_resolutions: list = [16, 8], dropout: float = 0, channel_mult=(1, 2, 4, 8),
                 conv_resample: bool = True, dims: int = 2, num_classes=None, cond_fn: Optional[nn.Module] = None,
                 use_checkpoint: bool = False, use_fp16: bool = False, num_heads: int = 1, num_head_channels: int = -1,
                 num_heads_upsample: int = -1, use_scale_shift_norm: bool = False, resblock_updown: bool = False,
                 use_new_attention_order: bool = False, activation: nn.Module = nn.SiLU(), compute_dtype="bf16"):
        super().__init__()
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads

        self.data_shape = data_shape
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.dims = dims
        self.use_scale_shift_norm = use_scale_shift_norm
        self.compute_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32, "f32": torch.float32}.get(
            compute_dtype, compute_dtype)

        embedding_dim = model_channels * 4
        if isinstance(activation, str):
            activation = registry.get("activations", activation)()
        self.act_code = activation_code(activation)
        # key of the dropout masks (nn.Dropout(p) of every ResBlock, unet_v2.py:239, active in training mode): Philox, one stream per
        # block; the reference draws from torch's global generator instead - the masks differ, their distribution does not
        self.dropout_seed = int(os.environ.get("RHO_SEED", "777")) + int(os.environ.get("RANK", "0"))

        self.time_embed = nn.Sequential(nn.Linear(model_channels, embedding_dim), activation,
                                        nn.Linear(embedding_dim, embedding_dim))
        self.cond_fn = cond_fn
        if self.num_classes is not None:
            self.label_emb = None

        self.resblock_updown = resblock_updown

        def res(cin, cout, **updown):
            return ResBlock(cin, embedding_dim, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint,
                            use_scale_shift_norm=use_scale_shift_norm, activation=activation, **updown)

        def attn(c, heads):
            return AttentionBlock(c, use_checkpoint=use_checkpoint, num_heads=heads, num_head_channels=num_head_channels,
                                  use_new_attention_order=use_new_attention_order)

        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, ch, 3, padding=1))])
        self._feature_size = ch
        input_block_chans = [ch]
        ds = 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, int(mult * model_channels))]
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                self._feature_size += ch
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(
                    res(ch, ch, down=True) if resblock_updown else Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                input_block_chans.append(ch)
                ds *= 2
                self._feature_size += ch

        self.middle_block = TimestepEmbedSequential(res(ch, ch), attn(ch, num_heads), res(ch, ch))
        self._feature_size += ch

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = input_block_chans.pop()
                layers = [res(ch + ich, int(model_channels * mult))]
                ch = int(model_channels * mult)
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(res(ch, ch, up=True) if resblock_updown else Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
                self._feature_size += ch

        self.out = nn.Sequential(normalization(ch), activation,
                                 zero_module(conv_nd(dims, input_ch, out_channels, 3, padding=1)))
        self._engines = {}

    # ------------------------------------------------------------------ HIP engine plumbing
    def __deepcopy__(self, memo):
        """Copies (EMA shadow models, DDP replicas) carry parameters and configuration only: engines hold plans keyed by the
        identity of THIS module tree, ctypes descriptors and raw device pointers, and are rebuilt by the copy on first use."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_engines", "grad_hooks"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new.__dict__["_engines"] = {}
        return new

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engines"] = {}
        state.pop("grad_hooks", None)
        return state

    def engine(self, dtype: Optional[torch.dtype] = None):
        from ..engine.unet_engine import UNetEngine
        dtype = dtype or self.compute_dtype
        eng = self._engines.get(dtype)
        if eng is None:
            eng = self._engines[dtype] = UNetEngine(self, dtype)
        return eng

    def set_compute_dtype(self, dtype) -> "UNet":
        self.compute_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}.get(dtype, dtype)
        return self

    def forward(self, x, timesteps, y=None):
        """x [N, C, *spatial] float32, timesteps [N] int64, y labels (see unet_v2.py:685-732)."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        if torch.is_grad_enabled() and self.training:
            anchor = next((p for p in self.parameters() if p.requires_grad), None)
            if anchor is not None:
                from ..autograd import UNetFunction
                return UNetFunction.apply(x, anchor, self.engine(), timesteps, y, getattr(self, "grad_hooks", None))
        # the engine returns its (reused) output buffer; hand the caller a tensor of its own
        out = self.engine().forward(x, timesteps, y).clone()
        if y is not None and not torch.cuda.is_current_stream_capturing():
            # a stand-alone inference call with labels fails like the reference's lookup (conditioning.py:132) does: at once.
            # (the training path and the sampling loop poll the same sticky flag at their own sync points: DDPM._check_nan,
            #  the end of reverse_process)
            self.check_errors()
        return out

    def check_errors(self) -> None:
        """Host poll of the device-side error flags of every engine of this model (one synchronisation per engine): raises
        IndexError for a label value outside the parameter space (rho_multi_embed, the reference's conditioning.py:132)."""
        for eng in self._engines.values():
            eng.check_errors()
