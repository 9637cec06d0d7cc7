"""Lowering of a ``UNet`` module tree (models/unet_v2.py) to flat lists of HIP kernel launches,
forward AND backward.

Reference semantics: UNet.forward, rho_diffusion/models/unet_v2.py:685-732, with ResBlock._forward
(:273-293), AttentionBlock._forward (:336-342), Downsample / Upsample (:103-169) and the final
GroupNorm-SiLU-conv (:679-683); the backward is what torch autograd derives for those ops.

Forward, per ResBlock:
    gn_partial, gn_finalize            statistics + folded (GroupNorm * FiLM) affine per (n, c)
    conv3   (prologue: affine+SiLU)    in_layers   [+ additive embedding in the epilogue]
    gn_partial, gn_finalize
    [conv1x1 skip]
    conv3   (prologue: affine+SiLU, epilogue: + skip)
torch.cat of the skip connections, nearest-upsample, strides, SiLU, FiLM and residual adds never
exist as separate passes over HBM.

Backward, per conv: bias gradient (channel sums of dY), weight gradient (rho_conv_nd_wgrad, input
activation recomputed in its loader), data gradient (the forward kernel on dY with flipped /
transposed weights) followed by the GroupNorm+FiLM+SiLU backward (reduce / finalize / apply).
Only pre-norm activations, GroupNorm statistics and the attention log-sum-exp are kept from the
forward; normalised / activated tensors and attention probabilities are recomputed (the reference
recomputes attention too, unet_v2.py:334).  Where a tensor has several consumers its gradient
buffer is aliased (identity skips, residuals) or accumulated in the producing kernel's epilogue;
the choice is made once, when the plan is built.

A plan (all descriptors + all buffers) is built once per (batch, spatial shape, mode) and replayed;
buffer addresses are stable.  PyTorch supplies memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import nn

from .. import hip
from ..hip import check, ptr
from . import ops

Tensor = torch.Tensor


class _ConvW:
    """A conv weight prepared for the kernels: forward [taps, coutp, cinp] and (training) data-gradient
    [taps, ceil32(cin), coutp] layouts in the engine dtype + padded fp32 bias."""

    def __init__(self, weight: nn.Parameter, bias: nn.Parameter, dtype, row_src: Optional[Tensor] = None):
        self.weight, self.bias_param, self.dtype = weight, bias, dtype
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        k = list(weight.shape[2:])
        while len(k) < 3:
            k.insert(0, 1)
        self.kernel = tuple(int(v) for v in k)
        self.taps = int(k[0] * k[1] * k[2])
        self._geometry()
        ck = ops.elem_chunk(dtype)
        self.cinp = ((self.cin + ck - 1) // ck) * ck
        self.coutp = ((self.cout + 31) // 32) * 32
        self.row_src = row_src
        dev = weight.device
        self.w = torch.empty(self.taps, self.coutp, self.cinp, dtype=dtype, device=dev)
        self.b = torch.zeros(self.coutp, dtype=torch.float32, device=dev)
        self.wd: Optional[Tensor] = None     # dgrad weights, allocated with the first training plan
        self.zero_bias: Optional[Tensor] = None
        self.wph: Optional[list] = None      # sub-pixel phase weights [(phase_hw, tensor)] of a conv behind a nearest x2 upsample
        self.wphd: Optional[list] = None     # ... and their data-gradient layouts (training plans)
        self.ws2: Optional[list] = None      # parity split of a stride-2 conv: forward taps per input parity
        self.ws2d: Optional[list] = None     # ... data-gradient taps per parity of dX
        self.refresh()

    def _geometry(self) -> None:          # subclasses re-interpret the parameter (see _StemAsGemm / _HeadAsGemm)
        pass

    def _source(self) -> Tensor:
        w = self.weight.detach()
        return w if w.is_contiguous() else w.contiguous()

    def _bias_source(self) -> Tensor:
        return self.bias_param.detach()

    def enable_dgrad(self) -> None:
        if self.wd is None:
            rows = ((self.cin + 31) // 32) * 32
            self.wd = torch.empty(self.taps, rows, self.coutp, dtype=self.dtype, device=self.w.device)
            self.zero_bias = torch.zeros(rows, dtype=torch.float32, device=self.w.device)
            self._refresh_dgrad()

    def enable_phases(self, up_hw, dgrad: bool = False) -> None:
        """The conv sits behind a nearest x2 upsample of the axes flagged in up_hw: one 2-tap weight set per output parity
        (and, for training plans, its data-gradient layout)."""
        hs = (1, 2) if up_hw[0] else (0,)
        ws = (1, 2) if up_hw[1] else (0,)
        if self.wph is None:
            self.wph = [((a, b), ops.prep_conv_weight_phase(self._source(), self.dtype, (a, b))) for a in hs for b in ws]
        if dgrad and self.wphd is None:
            self.wphd = [((a, b), ops.prep_conv_weight_phase(self._source(), self.dtype, (a, b), dgrad=True)) for a in hs for b in ws]
            if self.zero_bias is None:
                self.zero_bias = torch.zeros(((self.cin + 31) // 32) * 32, dtype=torch.float32, device=self.w.device)

    S2_FWD = {0: (1,), 1: (0, 2)}        # taps of the stride-2 forward on the even / odd input rows of a strided axis
    S2_BWD = {0: (1,), 1: (2, 0)}        # taps of its data gradient on the even / odd rows of dX

    def enable_s2(self, dgrad: bool = False) -> None:
        """3-D stride-(1, 2, 2) conv as stride-1 launches per parity (rho_prep_conv_weight_sel)."""
        if self.ws2 is None:
            self.ws2 = [((a, b), ops.prep_conv_weight_sel(self._source(), self.dtype, (self.S2_FWD[a], self.S2_FWD[b])))
                        for a in (0, 1) for b in (0, 1)]
            self.zero_b = torch.zeros(self.coutp, dtype=torch.float32, device=self.w.device)
        if dgrad and self.ws2d is None:
            self.ws2d = [((a, b), ops.prep_conv_weight_sel(self._source(), self.dtype, (self.S2_BWD[a], self.S2_BWD[b]), flip_d=True,
                                                           dgrad=True)) for a in (0, 1) for b in (0, 1)]
            if self.zero_bias is None:
                self.zero_bias = torch.zeros(((self.cin + 31) // 32) * 32, dtype=torch.float32, device=self.w.device)

    def _refresh_dgrad(self) -> None:
        w = self.weight.detach()
        w = w if w.is_contiguous() else w.contiguous()
        check(hip.lib().rho_prep_conv_weight_dgrad(ptr(w), ptr(self.wd), hip.dtype_code(self.dtype), self.cout, self.cin, self.taps,
                                                   self.wd.shape[1], self.wd.shape[2], ptr(self.row_src), hip.stream()),
              "rho_prep_conv_weight_dgrad")

    # batchable: every layout is a gather out of the parameter's own storage (rho_prep_batch reads the parameter directly)
    batchable = True

    def layout_signature(self) -> tuple:
        return (id(self), self.weight.data_ptr(), self.bias_param.data_ptr(), self.wd is not None, self.wph is not None,
                self.wphd is not None, self.ws2 is not None, self.ws2d is not None, self.weight.is_contiguous())

    def prep_into(self, table: "ops.PrepTable") -> bool:
        """Append this conv's prepared layouts (what ``refresh`` writes) to a rho_prep_batch table; False if it cannot be batched."""
        if not self.batchable or not self.weight.is_contiguous() or not self.bias_param.is_contiguous():
            return False
        w = self._source()                      # a view of the parameter (reshape of a contiguous tensor)
        if w.data_ptr() != self.weight.data_ptr():
            return False
        table.add_fwd(w, self.w, self.row_src)
        bsrc = self._bias_source()
        table.add_vec(bsrc, self.b, perm=self.row_src, n=(self.row_src.numel() if self.row_src is not None else bsrc.numel()))
        if self.wd is not None:
            table.add_dgrad(w, self.wd, self.row_src)
        for lst, dg in ((self.wph, False), (self.wphd, True)):
            for ph, t in (lst or []):
                table.add_phase(w, t, ph, dgrad=dg)
        for (a, b), t in (self.ws2 or []):
            table.add_sel(w, t, (self.S2_FWD[a], self.S2_FWD[b]))
        for (a, b), t in (self.ws2d or []):
            table.add_sel(w, t, (self.S2_BWD[a], self.S2_BWD[b]), flip_d=True, dgrad=True)
        return True

    def refresh(self) -> None:
        ops.prep_conv_weight(self._source(), self.dtype, self.coutp, self.cinp, self.row_src, out=self.w)
        b = self._bias_source()
        if self.row_src is not None:
            b = b[self.row_src.long()]           # gather (data movement only)
        self.b[: b.numel()].copy_(b)
        if self.wd is not None:
            self._refresh_dgrad()
        if self.wph is not None:
            for ph, t in self.wph:
                ops.prep_conv_weight_phase(self._source(), self.dtype, ph, out=t)
        if self.wphd is not None:
            for ph, t in self.wphd:
                ops.prep_conv_weight_phase(self._source(), self.dtype, ph, out=t, dgrad=True)
        if self.ws2 is not None:
            for (a, b), t in self.ws2:
                ops.prep_conv_weight_sel(self._source(), self.dtype, (self.S2_FWD[a], self.S2_FWD[b]), out=t)
        if self.ws2d is not None:
            for (a, b), t in self.ws2d:
                ops.prep_conv_weight_sel(self._source(), self.dtype, (self.S2_BWD[a], self.S2_BWD[b]), flip_d=True, dgrad=True, out=t)


class _StemAsGemm(_ConvW):
    """Stem conv with cin * taps <= 32 viewed as a 1x1x1 conv over the im2col operand (rho_im2col_taps): weight
    [cout, cin * taps] in the (ci, kd, kh, kw) order of ``weight.reshape``."""

    def __init__(self, weight: nn.Parameter, bias: nn.Parameter, dtype):
        self.taps3 = int(weight[0, 0].numel())
        self.kernel3 = tuple([1] * (3 - (weight.dim() - 2)) + [int(v) for v in weight.shape[2:]])
        super().__init__(weight, bias, dtype)

    def _geometry(self):
        self.kernel, self.taps = (1, 1, 1), 1
        self.cin = self.weight.shape[1] * self.taps3

    def _source(self) -> Tensor:
        return self.weight.detach().reshape(self.cout, self.cin, 1, 1, 1).contiguous()

    def enable_dgrad(self) -> None:
        """A forward-only re-reading of the parameter: no data-gradient layout (the stem has no data gradient at all)."""


class _HeadAsGemm(_ConvW):
    """Head conv with cout == 1 viewed as a 1x1x1 conv cin -> taps (rows = taps, padded to 32 output channels) whose
    result rho_tap_gather_sum folds over the taps; the bias is added there."""

    # its source is a transposed, zero-padded copy of the parameter, not a view: batched as a gather (RHO_PREP_VEC) through an
    # index table built once - w[0][r = tap][c] = weight[0][c][r]
    def prep_into(self, table: "ops.PrepTable") -> bool:
        if not self.weight.is_contiguous():
            return False
        if getattr(self, "_perm", None) is None:
            r = torch.arange(self.w.shape[1]).view(-1, 1)
            c = torch.arange(self.w.shape[2]).view(1, -1)
            idx = torch.where((r < self.taps3) & (c < self.cin), c * self.taps3 + r, torch.full_like(r + c, -1))
            self._perm = idx.reshape(-1).to(torch.int32).to(self.w.device)
        table.add_vec(self.weight.detach().reshape(-1), self.w, perm=self._perm, n=self.w.numel())
        return True                                  # (b stays zero: the bias is added by the kernel that folds the taps)



    def __init__(self, weight: nn.Parameter, bias: nn.Parameter, dtype):
        self.taps3 = int(weight[0, 0].numel())
        self.kernel3 = tuple([1] * (3 - (weight.dim() - 2)) + [int(v) for v in weight.shape[2:]])
        super().__init__(weight, bias, dtype)

    def _geometry(self):
        self.kernel, self.taps = (1, 1, 1), 1
        self.cout = 32

    def _source(self) -> Tensor:
        w = self.weight.detach()[0].reshape(self.cin, self.taps3).t()          # [taps, cin]
        full = torch.zeros(32, self.cin, 1, 1, 1, dtype=w.dtype, device=w.device)
        full[: self.taps3, :, 0, 0, 0] = w
        return full

    def _bias_source(self) -> Tensor:
        return torch.zeros(32, dtype=torch.float32, device=self.weight.device)

    def enable_dgrad(self) -> None:
        """A forward-only re-reading of the parameter ([1, C, taps] read as 32 rows x C): the generic data-gradient preparation
        would index the parameter with this geometry - 32 * C elements of a tensor that holds 27 * C (an out-of-bounds read that
        faulted once the parameter sat at the end of its allocation, round 4).  The head's data gradient has weights of its own
        (_HeadDgradW) or runs on the plain 3x3x3 form."""


class _HeadDgradW:
    """Data-gradient weights of a one-output-channel 3x3x3 head conv in the layout rho_stem_conv3d reads ([1][C][32], taps as the
    contraction): dact[pos][c] = sum_tap W[0][c][tap] dpred[pos - (tap - 1)] is that kernel run on dpred with the taps mirrored,
    w[0][c][t] = weight[0][c][26 - t]  (training plans of the bf16 engine, round 4)."""

    batchable = True

    def __init__(self, weight: nn.Parameter, dtype):
        self.weight = weight
        C_ = weight.shape[1]
        self.taps3 = int(weight[0, 0].numel())
        dev = weight.device
        self.w = torch.zeros(1, C_, 32, dtype=dtype, device=dev)
        self.zero_bias = torch.zeros(C_, dtype=torch.float32, device=dev)
        c = torch.arange(C_).view(-1, 1)
        t = torch.arange(32).view(1, -1)
        idx = torch.where(t < self.taps3, c * self.taps3 + (self.taps3 - 1 - t), torch.full_like(c + t, -1))
        self._perm = idx.reshape(-1).to(torch.int32).to(dev)
        self.refresh()

    def layout_signature(self) -> tuple:
        return (id(self), self.weight.data_ptr())

    def prep_into(self, table: "ops.PrepTable") -> bool:
        if not self.weight.is_contiguous():
            return False
        table.add_vec(self.weight.detach().reshape(-1), self.w, perm=self._perm, n=self.w.numel())
        return True

    def refresh(self) -> None:
        src = self.weight.detach().reshape(-1).float()
        g = torch.where(self._perm >= 0, src[self._perm.clamp(min=0).long()], torch.zeros((), device=src.device))   # (data movement only)
        self.w.copy_(g.view_as(self.w))


class _Pool:
    """Exact-size buffer pool for backward temporaries (emission order == stream order)."""

    def __init__(self, device):
        self.device = device
        self.free: Dict[tuple, List[Tensor]] = {}
        self.all: List[Tensor] = []

    def get(self, shape, dtype) -> Tensor:
        key = (int(torch.Size(shape).numel()), dtype)
        lst = self.free.get(key)
        if lst:
            return lst.pop().view(*shape)
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        self.all.append(t)
        return t

    def put(self, t: Tensor) -> None:
        self.free.setdefault((t.numel(), t.dtype), []).append(t)


class UNetEngine:
    def __init__(self, model: nn.Module, dtype: torch.dtype):
        self.model = model
        self.dtype = dtype
        p = next(model.parameters())
        hip.require_gpu(p, "UNet parameters")
        hip.load()
        self.device = p.device
        self.dims = model.dims
        self.mc = model.model_channels
        self.ssn = bool(model.use_scale_shift_norm)
        # activation code of the network (1 = SiLU: fused into the conv loaders; anything else runs in the materialising passes)
        self.act = int(getattr(model, "act_code", 1))
        self._plans: Dict[tuple, "_Plan"] = {}
        self._convs: List[_ConvW] = []
        self._aux_weights: list = []          # prepared layouts that are not convolutions of their own (see _head_dgrad)
        self._conv_of: Dict[int, _ConvW] = {}
        self._film_blocks: List[nn.Module] = []
        self._omega: Optional[Tensor] = None
        self._cond_dev = None      # device-side tables of the label embedding (MultiEmbeddings), built on first use
        self._param_version = -1
        self._ptr_sig = None
        self._prep_table = None
        self._prep_sig = None
        self._prep_eager: List[_ConvW] = []
        self._last_train_plan: Optional["_Plan"] = None
        with torch.inference_mode(False):
            self._collect()
            self.refresh_weights(force=True)

    # ------------------------------------------------------------------ weights
    def _conv(self, mod: nn.Module, row_src: Optional[Tensor] = None) -> _ConvW:
        key = id(mod)
        if key not in self._conv_of:
            cw = _ConvW(mod.weight, mod.bias, self.dtype, row_src)
            self._conv_of[key] = cw
            self._convs.append(cw)
        return self._conv_of[key]

    def _conv_as_gemm(self, mod: nn.Module, cls) -> _ConvW:
        """The stem / head convolution re-read as a 1x1x1 GEMM (inference plans of the bf16 engine)."""
        key = (id(mod), cls.__name__)
        if key not in self._conv_of:
            cw = cls(mod.weight, mod.bias, self.dtype)
            self._conv_of[key] = cw
            self._convs.append(cw)
        return self._conv_of[key]

    def _head_dgrad(self, mod: nn.Module) -> "_HeadDgradW":
        """Mirrored taps-as-contraction weights of the one-channel head conv (training plans: its data gradient runs on rho_stem_conv3d)."""
        key = (id(mod), "_HeadDgradW")
        if key not in self._conv_of:
            hw = _HeadDgradW(mod.weight, self.dtype)
            self._conv_of[key] = hw
            self._aux_weights.append(hw)
        return self._conv_of[key]

    def _qkv_row_src(self, blk) -> Tensor:
        """Row gather that brings the qkv projection to the canonical [Q heads | K heads | V heads]
        order the attention kernel reads (legacy order interleaves q,k,v per head, unet_v2.py:384)."""
        c, h = blk.channels, blk.num_heads
        ch = c // h
        idx = torch.arange(3 * c, dtype=torch.int64)
        if not blk.use_new_attention_order:
            part = idx // c            # 0 = q, 1 = k, 2 = v in the canonical layout
            head = (idx % c) // ch
            i = idx % ch
            idx = head * 3 * ch + part * ch + i
        return idx.to(torch.int32).to(self.device)

    def _collect(self) -> None:
        from ..models.unet_v2 import AttentionBlock, Downsample, ResBlock, Upsample
        m = self.model
        for mod in m.modules():
            if isinstance(mod, ResBlock):
                self._film_blocks.append(mod)
                self._conv(mod.in_layers[2])
                self._conv(mod.out_layers[3])
                if not isinstance(mod.skip_connection, nn.Identity):
                    self._conv(mod.skip_connection)
            elif isinstance(mod, AttentionBlock):
                self._conv(mod.qkv, self._qkv_row_src(mod))
                self._conv(mod.proj_out)
            elif isinstance(mod, Downsample):
                if mod.use_conv:
                    self._conv(mod.op)
            elif isinstance(mod, Upsample):
                if mod.use_conv:
                    self._conv(mod.conv)
        self._conv(m.input_blocks[0][0])
        self._conv(m.out[2])
        # FiLM / additive-embedding projections of all ResBlocks, batched into one GEMV launch
        self._film_off: Dict[int, int] = {}
        off = 0
        for blk in self._film_blocks:
            self._film_off[id(blk)] = off
            off += blk.emb_layers[1].weight.shape[0]
        self.film_total = off
        e = 4 * self.mc
        self.film_w = torch.empty(off, e, dtype=torch.float32, device=self.device)
        self.film_b = torch.empty(off, dtype=torch.float32, device=self.device)

    def _versions(self) -> int:
        return sum(p._version for p in self.model.parameters())

    def _pointer_signature(self) -> int:
        return hash(tuple(p.data_ptr() for p in self.model.parameters()))

    def refresh_weights(self, force: bool = False) -> None:
        """Re-run the weight preparation kernels if any parameter changed (optimizer step, load_state_dict);
        drop the plans if parameter storage moved (e.g. re-homed into an optimizer arena)."""
        sig = self._pointer_signature()
        if self._ptr_sig is not None and sig != self._ptr_sig:
            self._plans.clear()
            self._last_train_plan = None
            force = True
        self._ptr_sig = sig
        v = self._versions()
        if not force and v == self._param_version:
            return
        if os.environ.get("RHO_BATCH_PREP", "1") != "0":
            # ONE launch (rho_prep_batch) re-packs every layout of every conv, the padded biases and the FiLM matrix; the table is
            # rebuilt when a layout is added (first training plan, phases, parity splits) or parameter storage moves
            allw = self._convs + self._aux_weights
            lsig = tuple(cw.layout_signature() for cw in allw) + tuple(blk.emb_layers[1].weight.data_ptr() for blk in self._film_blocks)
            if self._prep_table is None or self._prep_sig != lsig:
                table = ops.PrepTable(self.device)
                self._prep_eager = [cw for cw in allw if not cw.prep_into(table)]
                off = 0
                for blk in self._film_blocks:
                    lin = blk.emb_layers[1]
                    n = lin.weight.shape[0]
                    table.add_vec(lin.weight.detach().reshape(-1), self.film_w[off:off + n].reshape(-1))
                    table.add_vec(lin.bias.detach(), self.film_b[off:off + n])
                    off += n
                self._prep_table, self._prep_sig = table, lsig
            self._prep_table.launch()
            for cw in self._prep_eager:
                cw.refresh()
            self._param_version = v
            return
        for cw in self._convs + self._aux_weights:
            cw.refresh()
        off = 0
        for blk in self._film_blocks:
            lin = blk.emb_layers[1]
            n = lin.weight.shape[0]
            self.film_w[off:off + n].copy_(lin.weight.detach())
            self.film_b[off:off + n].copy_(lin.bias.detach())
            off += n
        self._param_version = v

    def omega(self) -> Tensor:
        """Denominators wavelength^(2i / mc) of the timestep sinusoid (float32 [mc / 2]); the kernel evaluates sin / cos for any t."""
        if self._omega is None:
            self._omega = ops.sinusoid_frequencies(self.mc, 10000, self.device)
        return self._omega

    def err_flag(self) -> Tensor:
        """int32[1] device flag the kernels OR error bits into (bit 1: label not in the parameter space); polled by ``check_errors``."""
        if getattr(self, "_err_flag", None) is None:
            self._err_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        return self._err_flag

    def check_errors(self) -> None:
        """Host poll of the device error flag (one synchronisation: call it outside the hot loop)."""
        if getattr(self, "_err_flag", None) is not None:
            v = int(self._err_flag.item())
            if v & 2:
                self._err_flag.zero_()
                raise IndexError("MultiEmbeddings: a label value is not in the parameter space (conditioning.py:132)")

    def cond_device_tables(self):
        """Device-side description of ``model.cond_fn`` when it is a MultiEmbeddings (conditioning.py:31-139): the value lists of
        the parameter space concatenated as float32, their offsets, and a pointer array of the embedding tables.  None for any
        other cond_fn module (evaluated as given)."""
        from ..models.conditioning import MultiEmbeddings
        fn = getattr(self.model, "cond_fn", None)
        if not isinstance(fn, MultiEmbeddings) or fn.parameter_space is None or len(fn.embedding_layers) == 0:
            return None
        weights = [layer.weight for layer in fn.embedding_layers.values()]
        sig = tuple(w.data_ptr() for w in weights)
        if self._cond_dev is not None and self._cond_dev["sig"] == sig:
            return self._cond_dev
        keys = list(fn.embedding_layers.keys())
        if len(keys) > 16:
            return None
        vals, off = [], [0]
        for k in keys:
            v = torch.tensor(fn.parameter_space[k]).to(torch.float32)      # torch.tensor(list): the reference's conversion (:131)
            vals.append(v)
            off.append(off[-1] + v.numel())
        dev = self.device
        self._cond_dev = dict(sig=sig, nkeys=len(keys), weights=weights,
                              space=torch.cat(vals).to(dev).contiguous(),
                              key_off=torch.tensor(off, dtype=torch.int32, device=dev),
                              tables=torch.tensor(list(sig), dtype=torch.int64, device=dev))
        return self._cond_dev

    def param_order(self) -> List[nn.Parameter]:
        """Embedding-path parameters first (their gradients complete last), then every other parameter in
        forward order: backward finalises gradients from the tail of this list to its head, so contiguous
        ranges of an optimizer arena laid out this way are ready-made data-parallel all-reduce buckets."""
        emb = list(self.model.time_embed.parameters())
        for blk in self._film_blocks:
            emb.extend(blk.emb_layers[1].parameters())
        if getattr(self.model, "cond_fn", None) is not None:
            emb.extend(self.model.cond_fn.parameters())
        emb_ids = {id(p_) for p_ in emb}
        main = [p_ for p_ in self.model.parameters() if id(p_) not in emb_ids]
        return emb + main

    # ------------------------------------------------------------------ forward / backward
    def forward(self, x: Tensor, timesteps: Tensor, y: Optional[Tensor] = None,
                t_scalar_dev: Optional[Tensor] = None, train: bool = False) -> Tensor:
        hip.require_gpu(x, "x")
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        self.refresh_weights()
        key = (tuple(x.shape), y is not None, bool(train)) + self._plan_signature()
        plan = self._plans.get(key)
        if plan is None:
            with torch.inference_mode(False), torch.no_grad():     # plan buffers must stay ordinary tensors
                plan = self._plans[key] = _Plan(self, tuple(x.shape), y is not None, bool(train))
        if train:
            self._last_train_plan = plan
        return plan.run(x, timesteps, y, t_scalar_dev)

    # environment switches a plan reads while it is built (A/B knobs): part of the plan key, so flipping one rebuilds the plan
    _PLAN_ENV = ("RHO_TRAIN_MATERIALIZE", "RHO_MATERIALIZE_MIN_COUT", "RHO_PHASE_UPSAMPLE", "RHO_PHASE_UPSAMPLE_BWD", "RHO_PHASE_MIN_WGS",
                 "RHO_FOLD_SKIP", "RHO_FOLD_SKIP_TRAIN", "RHO_S2_SPLIT", "RHO_S2_SPLIT_BWD", "RHO_FUSE_GN_BWD", "RHO_GEMM_ENDS",
                 "RHO_DIRECT_ENDS", "RHO_CONV_SPLITK", "RHO_BATCH_PREP", "RHO_DW_ARENA", "RHO_FOLD_ADD", "RHO_FIN_BATCH_MB", "RHO_BWD_OVERLAP", "RHO_DIRECT_ENDS_TRAIN",
                 "RHO_FUSE_SKIP_DGRAD")

    def _plan_signature(self) -> tuple:
        """Everything besides (shape, labels, mode) that is baked into a plan when it is built: the per-ResBlock ``use_checkpoint``
        flags (layers.py:153-199 semantics, honoured per block), the library's deterministic switch and the A/B environment knobs.
        A change of any of them selects (builds) another plan instead of silently replaying the old one."""
        ck = tuple(bool(b.use_checkpoint) for b in self._film_blocks)
        env = tuple(os.environ.get(k) for k in self._PLAN_ENV)
        # nn.Dropout is active iff the module is in training mode (with or without autograd): part of the key where p > 0
        drop = bool(self.model.training) if any(float(getattr(b, "dropout", 0.0) or 0.0) > 0.0 for b in self._film_blocks) else None
        return (ck, ops.deterministic(), env, drop)

    def drop_plans(self, train_only: bool = False) -> None:
        """Release cached plans (and their device buffers): all of them, or only the training plans."""
        self._plans = {k: v for k, v in self._plans.items() if train_only and not k[2]}
        self._last_train_plan = None

    def backward(self, dpred: Tensor, on_ready: Optional[Callable[[List[nn.Parameter]], None]] = None) -> None:
        """Backward of the most recent ``forward(train=True)``: accumulates into ``p.grad`` of every
        parameter (allocated as zeros if missing).  ``on_ready(params)`` is called as groups of
        parameter gradients become final (used to overlap the data-parallel all-reduce)."""
        plan = self._last_train_plan
        if plan is None:
            raise hip.RhoHipError("backward() without a preceding forward(train=True)")
        with torch.no_grad():
            plan.run_backward(dpred, on_ready)


class _Plan:
    """All buffers + launch closures for one input shape (forward, and backward when train=True)."""

    def __init__(self, eng: UNetEngine, xshape: Tuple[int, ...], has_y: bool, train: bool):
        from ..models.unet_v2 import AttentionBlock, Downsample, ResBlock, Upsample
        self.eng = eng
        self.train = train
        self.materialize_act = os.environ.get("RHO_TRAIN_MATERIALIZE", "1") != "0"    # memory-for-time trade of training plans
        self.materialize_min_cout = int(os.environ.get("RHO_MATERIALIZE_MIN_COUT", "256"))
        # Upsample + conv as sub-pixel phases (A/B switch)
        self.phase_upsample = os.environ.get("RHO_PHASE_UPSAMPLE", "1") != "0"
        self.phase_upsample_bwd = os.environ.get("RHO_PHASE_UPSAMPLE_BWD", "1") != "0"
        self.phase_min_wgs = int(os.environ.get("RHO_PHASE_MIN_WGS", "256"))
        # bf16 engine: the ResBlock's 1x1x1 skip convolution inside its out-conv's forward launch (A/B switches)
        # (training plans: the forward launch only - backward keeps the skip branch's own data / weight-gradient launches)
        self.fold_skip = (eng.dtype == torch.bfloat16 and os.environ.get("RHO_FOLD_SKIP", "1") != "0"
                          and (not train or os.environ.get("RHO_FOLD_SKIP_TRAIN", "1") != "0"))
        self.s2_split = os.environ.get("RHO_S2_SPLIT", "1") != "0"
        self.s2_split_bwd = os.environ.get("RHO_S2_SPLIT_BWD", "1") != "0"
        # backward: GroupNorm's reductions (sum dz, sum dz * x) in the epilogue of the dgrad launch that produces dz (A/B switch)
        # - from RHO_FUSE_GN_BWD channels up (0 = never): on the 64-channel layers the extra epilogue VALU work (one sigmoid per
        # element) costs the issue-bound narrow tiles more than the separate reduce pass it replaces
        self.fuse_gn_bwd = int(os.environ.get("RHO_FUSE_GN_BWD", "128"))
        # backward of a ResBlock with a 1x1x1 skip convolution: the skip's data gradient and the GroupNorm backward apply of the
        # in-conv path write the same dX - one launch (rho_conv_desc.gna_*) instead of a data-gradient launch plus an apply pass that
        # re-reads and re-writes it (A/B switch)
        self.fuse_skip_dgrad = os.environ.get("RHO_FUSE_SKIP_DGRAD", "1") != "0"
        m = eng.model
        dt = eng.dtype
        dtc = hip.dtype_code(dt)
        dev = eng.device
        self.ops: List[Callable[[int], int]] = []
        self.info: List[dict] = []     # per launch: kind, algorithmic flops / bytes (for bench roofline)
        self.keep: List[object] = []   # descriptors / tensors referenced by raw pointers
        self.tstats: Dict[int, Tuple[Tensor, int]] = {}   # conv output data_ptr -> (fused statistics buffer, tiles per sample)
        self.nodes: List[dict] = []
        self.fwd_descs: List[object] = []          # rho_conv_desc of every forward / data-gradient launch (variants())
        self.wgrad_descs: List[tuple] = []         # (forward-shaped descriptor, dY row width) of every weight-gradient launch
        self.cond_src = None
        # dropout (nn.Dropout(p) of ResBlock.out_layers, unet_v2.py:239; active iff the model is in training mode): Philox masks in the
        # materialising pass, regenerated in backward from (seed of the block, a device counter advanced once per forward)
        self.drop_active = bool(eng.model.training) and any(float(getattr(b, "dropout", 0.0) or 0.0) > 0.0 for b in eng._film_blocks)
        self.drop_ctr = torch.zeros(1, dtype=torch.int64, device=eng.device) if self.drop_active else None
        self.drop_delta = 0
        self.drop_nodes: List[dict] = []          # (test aid) block, p, seed, activated-tensor shape of every dropout site
        L = hip.lib()
        self.L = L
        B = xshape[0]
        self.B = B
        self.xshape = xshape
        D, H, W = ops.spatial5(xshape[2:])
        dims = eng.dims
        e = 4 * eng.mc
        esz = 2 if dt == torch.bfloat16 else 4

        def buf(*shape, dtype=dt):
            t = torch.empty(*shape, dtype=dtype, device=dev)
            self.keep.append(t)
            return t

        scratch_of: Dict[tuple, Tensor] = {}

        def scratch(*shape, dtype=dt):
            """A buffer that is dead once the launch after its producer has run (the materialised activated input of ONE conv):
            shared by every request of the same size - launches of a plan are stream-ordered, also inside a captured graph."""
            key = (int(torch.Size(shape).numel()), dtype)
            if key not in scratch_of:
                scratch_of[key] = buf(key[0], dtype=dtype)
            return scratch_of[key].view(*shape)

        # ---- embedding chain: table gather -> Linear -> (SiLU) Linear (+cond) -> (SiLU) batched FiLM GEMV
        self.t_in = buf(B, dtype=torch.int64)
        self.sin_in = buf(B, eng.mc, dtype=torch.float32)
        self.emb_h = buf(B, e, dtype=torch.float32)     # PRE-activation of time_embed[0]; the consumer applies SiLU
        self.emb = buf(B, e, dtype=torch.float32)
        self.cond = buf(B, e, dtype=torch.float32) if has_y else None
        self.film = buf(B, max(eng.film_total, 1), dtype=torch.float32)
        te0, te2 = m.time_embed[0], m.time_embed[2]

        def op_linear(xt, w, b, add, out, act_in, act_out):
            Bn, K = xt.shape
            O = w.shape[0]
            args = (ptr(xt), ptr(w), ptr(b), ptr(add), ptr(out), Bn, K, O, int(act_in), int(act_out))
            self.keep.append((xt, w, b, add, out))
            self.ops.append(lambda s, a=args: L.rho_linear(*a, s))
            self.info.append(dict(kind="linear", flops=2.0 * Bn * K * O, bytes=4.0 * (O * K + Bn * (K + O))))

        # sinusoid + Linear + SiLU + Linear (+ label embedding) run as ONE launch at the head of run() (rho_timestep_embed: its
        # timestep pointer changes per call); sin_in / emb_h are kept for the backward
        self.cond_idx = buf(B, 16, dtype=torch.int32) if has_y else None
        if eng.film_total:
            op_linear(self.emb, eng.film_w, eng.film_b, None, self.film, eng.act, False)

        # ---- helpers that append launches
        def gn(x1, x2, norm, film_blk=None):
            N = x1.shape[0]
            c1 = x1.shape[-1]
            c2 = x2.shape[-1] if x2 is not None else 0
            S = x1.numel() // (N * c1)
            Cc = c1 + c2
            nblk = ops.gn_nblk(S)
            part = buf(N * nblk * (Cc // 8) * 16, dtype=torch.float32)
            a = buf(N, Cc, dtype=torch.float32)
            b = buf(N, Cc, dtype=torch.float32)
            st = buf(N, 32, 2, dtype=torch.float32)
            scale = shift = None
            stride = 0
            off = None
            if film_blk is not None:
                off = eng._film_off[id(film_blk)]
                scale = self.film.data_ptr() + 4 * off
                shift = self.film.data_ptr() + 4 * (off + Cc)
                stride = self.film.shape[1]
            # per-source partial sums: the producing convolution's fused epilogue statistics when it has them
            # (fmt 1, no extra read of the activation), else one rho_gn_partial pass over that source (fmt 0)
            srcs = []
            for xi, ci in ((x1, c1), (x2, c2)):
                if xi is None:
                    continue
                ts = self.tstats.get(xi.data_ptr())
                if ts is not None:
                    srcs.append((ptr(ts[0]), 1, ts[1], ci))
                else:
                    nb_i = ops.gn_nblk(S)
                    part_i = part if len(srcs) == 0 and x2 is None else buf(N * nb_i * (ci // 8) * 16, dtype=torch.float32)
                    a1 = (ptr(xi), ci, None, 0, dtc, N, S, ptr(part_i))
                    self.ops.append(lambda s, a=a1: L.rho_gn_partial(*a, s))
                    self.info.append(dict(kind="gn_partial", flops=3.0 * N * S * ci, bytes=float(esz) * N * S * ci))
                    srcs.append((ptr(part_i), 0, nb_i, ci))
            s1 = srcs[0]
            s2 = srcs[1] if len(srcs) > 1 else (None, 0, 0, 0)
            a2 = (s1[0], s1[1], s1[2], s1[3], s2[0], s2[1], s2[2], s2[3], N, S, ptr(norm.weight), ptr(norm.bias), scale, shift,
                  stride, ptr(st), ptr(a), ptr(b))
            self.ops.append(lambda s, a=a2: L.rho_gn_finalize2(*a, s))
            self.info.append(dict(kind="gn_finalize", flops=0.0, bytes=4.0 * N * Cc * 4))
            return dict(x1=x1, x2=x2, norm=norm, film_off=off, a=a, b=b, st=st, part=part, N=N, S=S, C=Cc, nblk=nblk)

        def conv(x1, x2, cw, *, stride_hw=(1, 1), up_hw=(0, 0), pre=None, pre_silu=False, res=None, res_add_off=None,
                 split=None, y2_dtype=None, stem=False, want_stats=True, ckpt=False, fold_skip=None, node_res=None, drop=None):
            cout = cw.cout
            split_ = cout if split is None else split
            N, Do, Ho, Wo = ops.conv_out_shape(x1.shape, cw.kernel, stride_hw, up_hw)
            y = buf(N, Do, Ho, Wo, split_) if split_ > 0 else None
            y2 = buf(N, cout - split_, Do * Ho * Wo, dtype=y2_dtype or dt) if split_ < cout else None
            xact = None
            cx1, cx2, cpre = x1, x2, pre
            # 1x1x1 projections with many cout tiles (the attention qkv: 12 tiles of 128) redo the prologue per tile with
            # nothing to hide it under (probe: 0.70 ms with, 0.43 ms without, for 0.06 ms of materialising pass)
            wide_1x1 = cw.taps == 1 and cout >= 512
            # 3x3(x3) convs with >= 2 cout tiles of 128 (the 256- and 512-wide levels): every cout tile redoes GroupNorm + SiLU on the
            # halo tile it stages (2.5x the input per tile): 8.5 % of the launch against a 0.05 ms pass that applies it once
            # (tools/ab_conv.py, "+pre" rows; only where the tensor is small enough that the extra pass costs less than the prologue)
            wide_3x3 = (cw.taps > 1 and cout >= self.materialize_min_cout and dt == torch.bfloat16)
            # ``ckpt`` (a ResBlock built with use_checkpoint=True; reference: layers.py:153-199 re-runs the block in backward instead of
            # keeping its intermediates): the activated inputs of the block's convs are NOT kept - backward re-materialises them
            # into a recycled buffer (the recompute path of bias_and_wgrad), the forward conv applies GroupNorm + FiLM + SiLU in its
            # loader or, for the wide layers, from a scratch copy that the next conv overwrites
            keep_act = self.train and self.materialize_act and not ckpt
            # (an activation other than SiLU exists in the materialising pass only: the conv loaders know identity and SiLU)
            other_act = pre is not None and int(pre_silu) > 1
            if other_act and up_hw != (0, 0):
                raise hip.RhoHipError("internal: a normalised conv behind an upsample with a non-SiLU activation")
            if drop is not None and (pre is None or up_hw != (0, 0)):
                raise hip.RhoHipError("internal: dropout on a conv without a materialisable normalised input")
            if pre is not None and up_hw == (0, 0) and (keep_act or wide_1x1 or wide_3x3 or other_act or drop is not None):
                # training: the activated input act(a*x+b) is needed twice (this conv, its weight gradient) and the conv
                # loader would recompute it 2.3x (halo) per cout tile: materialise it once (one HBM-rate pass, kept for
                # backward: +1 activation-sized buffer per normalised conv, 38 GB at c3) and feed both from it
                c1_ = x1.shape[-1]
                c2_ = x2.shape[-1] if x2 is not None else 0
                xact = (buf if keep_act else scratch)(*x1.shape[:4], c1_ + c2_)
                Sx = x1.shape[1] * x1.shape[2] * x1.shape[3]
                ga = (ptr(x1), c1_, ptr(x2), c2_, dtc, x1.shape[0], Sx, ptr(pre["a"]), ptr(pre["b"]), int(pre_silu), ptr(xact))
                if drop is not None:
                    gd = ga + (float(drop[0]), int(drop[1]), ptr(self.drop_ctr))
                    self.ops.append(lambda s, a=gd: L.rho_gn_apply_drop(*a, s))
                    self.drop_delta = max(self.drop_delta, (xact.numel() + 3) // 4)
                    self.drop_nodes.append(dict(blk=drop[2], p=float(drop[0]), seed=int(drop[1]), shape=tuple(xact.shape)))
                else:
                    self.ops.append(lambda s, a=ga: L.rho_gn_apply(*a, s))
                self.info.append(dict(kind="gn_apply", flops=0.0, bytes=2.0 * esz * xact.numel()))
                cx1, cx2, cpre = xact, None, None
            # A conv behind a nearest x2 upsample as one 2-tap launch per output parity on the SOURCE tensor (rho_conv_desc.ph_h):
            # 12 / 27 of the multiply-adds in 3-D, same result up to the rounding of the summed weights.
            # (only where each phase launch still fills the chip: on the small 2-D grids of c1 four launches of a few workgroups
            #  each are slower than one - 20.2 -> 21.5 ms per step there)
            n_ph = (2 if up_hw[0] else 1) * (2 if up_hw[1] else 1)
            wgs_per_phase = (N * Do * Ho * Wo // n_ph // 256) * max(1, cw.coutp // 128)
            phased = (self.phase_upsample and up_hw != (0, 0) and cpre is None and cx2 is None and split_ == cout and res is None
                      and res_add_off is None and all(cw.kernel[1 + i] == 3 for i in range(2) if up_hw[i])
                      and wgs_per_phase >= self.phase_min_wgs)
            # Downsample's stride-(1, 2, 2) conv as four stride-1 launches, one per input parity, accumulated in place: no 2x halo
            # per strided tile (the strided loader ran 380 - 870 TF/s), same multiply-adds
            s2 = (self.s2_split and tuple(stride_hw) == (2, 2) and tuple(cw.kernel) == (3, 3, 3) and up_hw == (0, 0) and cpre is None
                  and cx2 is None and split_ == cout and res is None and res_add_off is None and x1.shape[2] % 2 == 0
                  and x1.shape[3] % 2 == 0 and (N * Do * Ho * Wo // 256) * max(1, cw.coutp // 128) >= self.phase_min_wgs)
            if s2:
                cw.enable_s2(dgrad=self.train)
                descs = [ops.make_conv_desc(cx1, None, wt, cw.b if i == 0 else cw.zero_b, kernel=(3, len(cw.S2_FWD[a]), len(cw.S2_FWD[b])),
                                            cout=cout, split=split_, y=y, y2=None, res=y if i > 0 else None, phase_dgrad_hw=(a + 1, b + 1))
                         for i, ((a, b), wt) in enumerate(cw.ws2)]
            elif phased:
                cw.enable_phases(up_hw, dgrad=self.train)
                descs = [ops.make_conv_desc(cx1, None, wt, cw.b, kernel=(cw.kernel[0], 2 if ph[0] else cw.kernel[1], 2 if ph[1] else cw.kernel[2]),
                                            cout=cout, split=split_, y=y, y2=None, phase_hw=ph) for ph, wt in cw.wph]
            else:
                d = ops.make_conv_desc(cx1, cx2, cw.w, cw.b, kernel=cw.kernel, cout=cout, split=split_, y=y, y2=y2,
                                       stride_hw=stride_hw, up_hw=up_hw, pre_a=cpre["a"] if cpre else None,
                                       pre_b=cpre["b"] if cpre else None, pre_silu=pre_silu if cpre else False, res=res, res_add=None,
                                       skip=fold_skip)
                if res_add_off is not None:
                    d.res_add = self.film.data_ptr() + 4 * res_add_off
                    d.res_add_stride = self.film.shape[1]
                descs = [d]
            if y is not None and split_ == cout and want_stats:
                # GroupNorm statistics of the output ride along in the epilogue where the geometry allows it
                tiles = int(L.rho_conv_stats_tiles(C.byref(descs[0])))       # (phases: all launches of this output together)
                tiles = int(L.rho_conv_stats_tiles(C.byref(descs[-1]))) if s2 else tiles
                if tiles > 0:
                    sbuf = buf(N * tiles * 2 * cout, dtype=torch.float32)
                    for d in (descs[-1:] if s2 else descs):      # (parity split: the last launch stores the final values)
                        d.stats = sbuf.data_ptr()
                    self.tstats[y.data_ptr()] = (sbuf, tiles)
            npos_out = N * Do * Ho * Wo
            npos_in = x1.numel() // x1.shape[-1]
            for d in descs:
                self.keep.append(d)
                self.fwd_descs.append(d)
                self.ops.append(lambda s, d=d: L.rho_conv_nd_fwd(C.byref(d), s))
                taps_run = d.kd * d.kh * d.kw
                self.info.append(dict(
                    kind="conv3" if cw.taps > 1 else "conv1", taps=cw.taps, cin=cw.cin, cout=cout,
                    positions=npos_out if s2 else npos_out // len(descs),
                    flops=(2.0 * npos_out * cout * cw.cin * taps_run if s2 else
                           2.0 * npos_out * cout * cw.cin * cw.taps / len(descs)),         # algorithmic (unpadded) MACs * 2
                    executed_flops=2.0 * npos_out * cout * cw.cin * taps_run / (1 if s2 else len(descs)),
                    bytes=float(esz) * (npos_in * cw.cin + npos_out * cout * (2 if res is not None else 1) / len(descs)
                                        + taps_run * cout * cw.cin)))
            # (node_res: the backward's view of a folded skip - the gradient of this output also belongs to the skip branch's node)
            self.nodes.append(dict(k="conv", cw=cw, x1=x1, x2=x2, y=y, y2=y2, stride_hw=stride_hw, up_hw=up_hw, pre=pre,
                                   pre_silu=pre_silu, res=res if node_res is None else node_res, res_add_off=res_add_off, stem=stem,
                                   out_dims=(N, Do, Ho, Wo),
                                   xact=xact if keep_act else None, phased=phased, s2=s2, drop=drop))
            return y, y2

        rs_hw = (1, 1) if dims >= 2 else (0, 1)      # axes a Down/Upsample touches: H and W (3-D: depth stays), 1-D: W only

        def resample(xt, mode):
            """avg_pool_nd (mode "avg") / nearest x2 (mode "up") of a channels-last tensor as its own pass: the conv-less
            Down/Upsample of conv_resample = False and the h_upd / x_upd of ResBlock(up / down) (unet_v2.py:122-131,165,221-224)."""
            N_, Dd, Hh, Ww, Cc = xt.shape
            if mode == "up":
                yt = buf(N_, Dd, Hh * 2 if rs_hw[0] else Hh, Ww * 2, Cc)
                a = (ptr(xt), ptr(yt), dtc, N_ * Dd, Hh, Ww, Cc, rs_hw[0], rs_hw[1])
                self.ops.append(lambda s, a=a: L.rho_upsample2x(*a, s))
            else:
                yt = buf(N_, Dd, Hh // 2 if rs_hw[0] else Hh, Ww // 2, Cc)
                a = (ptr(xt), ptr(yt), dtc, N_ * Dd, Hh, Ww, Cc, rs_hw[0], rs_hw[1])
                self.ops.append(lambda s, a=a: L.rho_avgpool2x(*a, s))
            self.info.append(dict(kind="resample", flops=0.0, bytes=float(esz) * (xt.numel() + yt.numel())))
            self.nodes.append(dict(k="resample", mode=mode, x=xt, y=yt))
            return yt

        def activated(x1, x2, pre, pre_silu):
            """act(a * concat(x1, x2) + b) materialised as a tensor of its own (needed when something other than a conv loader
            consumes it: the h_upd of an up / down ResBlock resamples AFTER GroupNorm + SiLU, unet_v2.py:277-281)."""
            c1_ = x1.shape[-1]
            c2_ = x2.shape[-1] if x2 is not None else 0
            yt = buf(*x1.shape[:4], c1_ + c2_)
            Sx = x1.shape[1] * x1.shape[2] * x1.shape[3]
            ga = (ptr(x1), c1_, ptr(x2), c2_, dtc, x1.shape[0], Sx, ptr(pre["a"]), ptr(pre["b"]), int(pre_silu), ptr(yt))
            self.ops.append(lambda s, a=ga: L.rho_gn_apply(*a, s))
            self.info.append(dict(kind="gn_apply", flops=0.0, bytes=2.0 * esz * yt.numel()))
            self.nodes.append(dict(k="act", x1=x1, x2=x2, pre=pre, pre_silu=pre_silu, y=yt))
            return yt

        def drop_of(blk):
            """(p, seed, block) of the block's nn.Dropout when it is active in this plan, else None: one Philox key per block."""
            p_ = float(getattr(blk, "dropout", 0.0) or 0.0)
            if not self.drop_active or p_ <= 0.0:
                return None
            idx = eng._film_blocks.index(blk)
            seed = (int(getattr(eng.model, "dropout_seed", 777)) + 0x9E3779B97F4A7C15 * (idx + 1)) & 0xFFFFFFFFFFFFFFFF
            return (p_, seed, blk)

        def resblock_updown(blk, h1, h2):
            from ..models.unet_v2 import Upsample as _Up
            mode = "up" if isinstance(blk.h_upd, _Up) else "avg"
            g1 = gn(h1, h2, blk.in_layers[0])
            hh = resample(activated(h1, h2, g1, eng.act), mode)
            x1p = resample(h1, mode)
            x2p = resample(h2, mode) if h2 is not None else None
            radd = None if blk.use_scale_shift_norm else eng._film_off[id(blk)]
            t1, _ = conv(hh, None, eng._conv(blk.in_layers[2]), res_add_off=radd)
            g2 = gn(t1, None, blk.out_layers[0], film_blk=blk if blk.use_scale_shift_norm else None)
            if isinstance(blk.skip_connection, nn.Identity):
                assert x2p is None
                sk = x1p
            else:
                sk, _ = conv(x1p, x2p, eng._conv(blk.skip_connection))
            out, _ = conv(t1, None, eng._conv(blk.out_layers[3]), pre=g2, pre_silu=eng.act, res=sk, ckpt=bool(blk.use_checkpoint), drop=drop_of(blk))
            return out

        def resblock(blk, h1, h2):
            if getattr(blk, "updown", False):
                return resblock_updown(blk, h1, h2)
            g1 = gn(h1, h2, blk.in_layers[0])
            radd = None if blk.use_scale_shift_norm else eng._film_off[id(blk)]
            ck = bool(blk.use_checkpoint)
            t1, _ = conv(h1, h2, eng._conv(blk.in_layers[2]), pre=g1, pre_silu=eng.act, res_add_off=radd, ckpt=ck)
            g2 = gn(t1, None, blk.out_layers[0], film_blk=blk if blk.use_scale_shift_norm else None)
            if isinstance(blk.skip_connection, nn.Identity):
                assert h2 is None
                sk = h1
            else:
                skw, ocw = eng._conv(blk.skip_connection), eng._conv(blk.out_layers[3])
                if self.fold_skip and skw.taps == 1 and ocw.taps == 27:
                    # skip_connection(x) + out_layers(h) (unet_v2.py:245-256,293) in ONE forward launch - the 1x1x1 skip is
                    # contracted into the out-conv's accumulators before its tap loop (rho_conv_desc.sk_*): no launch, no `sk`
                    # tensor written and read back as the residual.  Where the kernel has no such variant (rho_conv_variant says
                    # so: narrow / wide cout tiles, large halos) the two launches stay.
                    fs = (h1, h2, skw.w, skw.b)
                    probe = ops.make_conv_desc(t1, None, ocw.w, ocw.b, kernel=ocw.kernel, cout=ocw.cout, split=ocw.cout,
                                               y=t1, y2=None, skip=fs)
                    if L.rho_conv_variant(C.byref(probe), C.create_string_buffer(128), 128) == 0:
                        skd = None
                        if self.train:
                            # backward is the unfused graph: a node for the skip branch whose "output" is a key-only tensor; the
                            # out-conv's node names it as its residual, so its dY is aliased to the skip node exactly as before
                            skd = buf(8, dtype=torch.uint8)
                            N_, D_, H_, W_ = t1.shape[:4]
                            self.nodes.append(dict(k="conv", cw=skw, x1=h1, x2=h2, y=skd, y2=None, stride_hw=(1, 1), up_hw=(0, 0), pre=None,
                                                   pre_silu=False, res=None, res_add_off=None, stem=False, out_dims=(N_, D_, H_, W_),
                                                   xact=None, phased=False, s2=False))
                        out, _ = conv(t1, None, ocw, pre=g2, pre_silu=eng.act, ckpt=ck, fold_skip=fs, node_res=skd, drop=drop_of(blk))
                        # the launch's work = the 27-tap conv + the folded 1x1x1 (both algorithmic FLOPs of the reference's
                        # formulation); the 1x1x1 share is also reported on its own (bench: roofline.folded_conv1_flops_per_step)
                        fl = 2.0 * (t1.numel() // t1.shape[-1]) * ocw.cout * skw.cin
                        self.info[-1]["flops"] += fl
                        self.info[-1]["executed_flops"] += fl
                        self.info[-1]["folded_conv1_flops"] = fl
                        self.info[-1]["bytes"] += float(esz) * (t1.numel() // t1.shape[-1]) * skw.cin
                        return out
                sk, _ = conv(h1, h2, skw)
            out, _ = conv(t1, None, eng._conv(blk.out_layers[3]), pre=g2, pre_silu=eng.act, res=sk, ckpt=ck, drop=drop_of(blk))
            return out

        def attention(blk, xin):
            N, Dd, Hh, Ww, Cc = xin.shape
            T = Dd * Hh * Ww
            g = gn(xin, None, blk.norm)
            qk, vt = conv(xin, None, eng._conv(blk.qkv), pre=g, pre_silu=False, split=2 * Cc)
            ao = buf(N, Dd, Hh, Ww, Cc)
            lse = buf(N, blk.num_heads, T, dtype=torch.float32) if train else None
            args = (ptr(qk), ptr(vt), ptr(ao), ptr(lse), dtc, N, T, blk.num_heads, Cc // blk.num_heads)
            self.ops.append(lambda s, a=args: L.rho_attention_fwd(*a, s))
            self.info.append(dict(kind="attention", flops=4.0 * N * T * T * Cc, bytes=float(esz) * 4 * N * T * Cc))
            self.nodes.append(dict(k="attn", qk=qk, vt=vt, ao=ao, lse=lse, heads=blk.num_heads, N=N, T=T, C=Cc))
            out, _ = conv(ao, None, eng._conv(blk.proj_out), res=xin)
            return out

        def run_block(seq, h1, h2):
            for layer in seq:
                if isinstance(layer, ResBlock):
                    h1, h2 = resblock(layer, h1, h2), None
                elif isinstance(layer, AttentionBlock):
                    h1 = attention(layer, h1)
                elif isinstance(layer, Downsample):
                    if not layer.use_conv:
                        h1 = resample(h1, "avg")
                    else:
                        st = (2, 2) if dims >= 2 else (1, 2)
                        h1, _ = conv(h1, None, eng._conv(layer.op), stride_hw=st)
                elif isinstance(layer, Upsample):
                    if not layer.use_conv:
                        h1 = resample(h1, "up")
                    else:
                        up = (1, 1) if dims >= 2 else (0, 1)
                        h1, _ = conv(h1, None, eng._conv(layer.conv), up_hw=up)
                elif stem_direct is not None:  # the stem conv as one launch on the fp32 input (rho_stem_conv3d)
                    h1 = stem_direct()
                else:  # the stem conv
                    h1, _ = conv(h1, None, stem, stem=True)
            return h1

        # ---- the network
        # Inference plans of the bf16 engine run a 1-channel stem / head as 1x1x1 GEMMs (see rho_im2col_taps /
        # rho_tap_gather_sum: the 3x3x3 form pads the single channel to 32 and spends 31/32 of its matrix work on zeros).
        gemm_ends = (not train) and dt == torch.bfloat16 and os.environ.get("RHO_GEMM_ENDS", "1") != "0"
        stem = eng._conv(m.input_blocks[0][0])
        self.x_in = buf(*xshape, dtype=torch.float32)
        # 3-D, one input / output channel: each end is ONE launch with its intermediate in LDS (csrc/ends.hip; A/B switch) instead
        # of the GEMM form's two (im2col + GEMM, GEMM + tap gather)
        direct_ends = gemm_ends and dims == 3 and os.environ.get("RHO_DIRECT_ENDS", "1") != "0"
        # training plans (round 4): the same two forward launches; their backward = GEMM-shaped weight gradients against an im2col
        # of the one-channel operand (k_wgrad1) and, for the head's data gradient, rho_stem_conv3d on dpred with mirrored taps -
        # instead of 3x3x3 launches whose single channel is padded to 32 (31 / 32 of their matrix work on zeros)
        direct_ends_train = (train and dt == torch.bfloat16 and dims == 3 and os.environ.get("RHO_DIRECT_ENDS", "1") != "0"
                             and os.environ.get("RHO_DIRECT_ENDS_TRAIN", "1") != "0" and os.environ.get("RHO_DW_ARENA", "1") != "0")
        stem_direct = None
        if (direct_ends or direct_ends_train) and xshape[1] == 1 and tuple(stem.kernel) == (3, 3, 3) and stem.cout in (32, 64):
            sg = eng._conv_as_gemm(m.input_blocks[0][0], _StemAsGemm)

            def stem_direct():
                y = buf(B, D, H, W, sg.cout)
                tiles = ops.stem_conv3d_tiles(D, H, W)
                sbuf = buf(B * tiles * 2 * sg.cout, dtype=torch.float32)
                a = (ptr(self.x_in), ptr(sg.w), ptr(sg.b), ptr(y), ptr(sbuf), B, D, H, W, sg.cout)
                self.ops.append(lambda s_, a=a: L.rho_stem_conv3d(*a, s_))
                self.tstats[y.data_ptr()] = (sbuf, tiles)
                npos = B * D * H * W
                self.info.append(dict(kind="stem", flops=2.0 * npos * sg.cout * 27, bytes=4.0 * npos + float(esz) * npos * sg.cout))
                if train:
                    self.nodes.append(dict(k="stem_direct", y=y, cw=stem, out_dims=(B, D, H, W)))
                return y
            self.x_cl = self.x_in
        elif gemm_ends and stem.taps > 1 and stem.cin * stem.taps <= 32:
            stem = eng._conv_as_gemm(m.input_blocks[0][0], _StemAsGemm)
            self.x_cl = buf(B, D, H, W, stem.cinp)
            pk = (ptr(self.x_in), ptr(self.x_cl), dtc, B, xshape[1], D, H, W) + stem.kernel3 + (stem.cinp,)
            self.ops.append(lambda s, a=pk: L.rho_im2col_taps(*a, s))
        else:
            self.x_cl = buf(B, D, H, W, stem.cinp)
            pk = (ptr(self.x_in), ptr(self.x_cl), dtc, B, xshape[1], D * H * W, stem.cinp)
            self.ops.append(lambda s, a=pk: L.rho_pack_input(*a, s))
        if stem_direct is None:
            self.info.append(dict(kind="pack", flops=0.0, bytes=4.0 * B * xshape[1] * D * H * W + 2.0 * B * D * H * W * stem.cinp))

        hs = []
        h = self.x_cl
        for blk in m.input_blocks:
            h = run_block(blk, h, None)
            hs.append(h)
        h = run_block(m.middle_block, h, None)
        for blk in m.output_blocks:
            h = run_block(blk, h, hs.pop())
        g = gn(h, None, m.out[0])
        head = eng._conv(m.out[2])
        if ((direct_ends or direct_ends_train) and tuple(head.kernel) == (3, 3, 3) and head.cout == 1
                and h.shape[-1] in ((32, 64) if train else (32, 64, 96, 128))):      # (training: dpred -> dact runs on rho_stem_conv3d)
            hg = eng._conv_as_gemm(m.out[2], _HeadAsGemm)
            y2 = buf(B, 1, D * H * W, dtype=torch.float32)
            npos = B * D * H * W
            if train or eng.act != 1:
                # training: the activated input is kept (the head's weight gradient contracts it with the im2col of dpred);
                # a non-SiLU activation: applied by the materialising pass (the head kernel's own prologue knows SiLU only)
                xact = (buf if train else scratch)(B, D, H, W, h.shape[-1])
                ga0 = (ptr(h), h.shape[-1], None, 0, dtc, B, D * H * W, ptr(g["a"]), ptr(g["b"]), eng.act, ptr(xact))
                self.ops.append(lambda s, a=ga0: L.rho_gn_apply(*a, s))
                self.info.append(dict(kind="gn_apply", flops=0.0, bytes=2.0 * esz * xact.numel()))
                ga = (ptr(xact), None, None, 0, ptr(hg.w), ptr(m.out[2].bias), ptr(y2), B, D, H, W, h.shape[-1])
                if train:
                    self.nodes.append(dict(k="head_direct", x=h, pre=g, xact=xact, cw=head, y2=y2, out_dims=(B, D, H, W)))
            else:
                ga = (ptr(h), ptr(g["a"]), ptr(g["b"]), 1, ptr(hg.w), ptr(m.out[2].bias), ptr(y2), B, D, H, W, h.shape[-1])
            self.ops.append(lambda s, a=ga: L.rho_head_conv3d(*a, s))
            self.keep.append(g)
            self.info.append(dict(kind="head", flops=2.0 * npos * h.shape[-1] * 27, bytes=float(esz) * npos * h.shape[-1] + 4.0 * npos))
        elif gemm_ends and head.taps > 1 and head.cout == 1:
            hg = eng._conv_as_gemm(m.out[2], _HeadAsGemm)
            tt, _ = conv(h, None, hg, pre=g, pre_silu=eng.act, want_stats=False)          # [B, D, H, W, 32]: one column per tap
            y2 = buf(B, 1, D * H * W, dtype=torch.float32)
            ga = (ptr(tt), dtc, B, D, H, W) + hg.kernel3 + (hg.coutp, ptr(m.out[2].bias), ptr(y2))
            self.ops.append(lambda s, a=ga: L.rho_tap_gather_sum(*a, s))
            self.info.append(dict(kind="tap_sum", flops=0.0, bytes=2.0 * tt.numel() + 4.0 * y2.numel()))
        else:
            _, y2 = conv(h, None, head, pre=g, pre_silu=eng.act, split=0, y2_dtype=torch.float32)
        self.out = y2.view(B, m.out_channels, *xshape[2:])

        self.bwd: List[Callable[[int], int]] = []
        self.bwd_info: List[dict] = []
        self.bwd_marks: List[Tuple[int, List[nn.Parameter]]] = []   # after bwd[:i] these parameters' gradients are final
        if train:
            self._build_backward()
        # k-split of the small-grid 2-D / 1-D launches (rho_conv_desc.ws): one workspace per plan, shared by its ordered launches
        ws = ops.attach_conv_workspace(self.fwd_descs, dev)
        if ws is not None:
            self.keep.append(ws)

    # ------------------------------------------------------------------ backward construction
    def _build_backward(self) -> None:
        eng, L = self.eng, self.L
        m = eng.model
        dt = eng.dtype
        dtc = hip.dtype_code(dt)
        dev = eng.device
        B = self.B
        pool = _Pool(dev)
        self.keep.append(pool)
        for cw in eng._convs:
            cw.enable_dgrad()
        G: Dict[int, Tensor] = {}        # activation data_ptr -> gradient buffer
        written = set()
        bw, binfo = self.bwd, self.bwd_info
        esz = 2 if dt == torch.bfloat16 else 4

        def key(t: Tensor) -> int:
            return t.data_ptr()

        # ---- overlap of the HBM-bound GroupNorm backward with the weight gradient (round 4; RHO_BWD_OVERLAP=0: one stream).  Per
        # normalised conv the order on the launch stream was  wgrad | dgrad | gn_bwd reduce / finalize / apply  - three kernels that
        # cannot share the chip (two matrix-bound, one at HBM rate: 24 ms of passes per c3 step with the matrix cores idle).  The
        # weight gradient is off the critical path (it needs dY only), so it now goes to a SIDE stream right after the data gradient:
        #     main:  dgrad ............ | gn_bwd reduce, finalize, apply | (wait) next node
        #     side:                     | wgrad ....................... |
        # one wave per SIMD of k_wgrad leaves registers and wave slots for the elementwise passes, which run in its shadow.  Only
        # this pair overlaps: the next node's launches wait for the side stream (two matrix-bound kernels sharing the CUs cost more in
        # L2 locality than they return - measured in round 3).  Pool buffers released by the weight-gradient path are recycled after
        # that wait only.
        # MEASURED (same-box A/B, c3): 406.3 / 402.1 ms per step with the side stream against 398.5 / 400.0 without - the passes do not
        # hide (the weight gradient slows by as much as they take), so this stays OFF; the switch remains for other shapes.
        overlap = os.environ.get("RHO_BWD_OVERLAP", "0") != "0"
        self._side = torch.cuda.Stream(device=dev) if overlap else None
        self._overlap_on = True                      # profile() turns it off: per-launch timings need the serial order
        defer: Dict[str, object] = {"on": False, "ops": [], "puts": []}

        def emit(fn, kind, flops=0.0, nbytes=0.0, **shape):
            info = dict(kind=kind, flops=flops, bytes=nbytes, **shape)
            if defer["on"]:
                defer["ops"].append((fn, info))
                return
            bw.append(fn)
            binfo.append(info)

        def wput(t: Tensor):
            """pool.put for buffers the weight-gradient path read: held back while that path is being deferred to the side stream."""
            if defer["on"]:
                defer["puts"].append(t)
            else:
                pool.put(t)

        def flush_side():
            """Emit the deferred weight-gradient launches as side-stream launches (called right after the data-gradient launch)."""
            for fn, info in defer["ops"]:
                ev = torch.cuda.Event()

                def run(s, fn=fn, ev=ev):
                    if not self._overlap_on:
                        return fn(s)
                    ev.record(torch.cuda.current_stream())
                    self._side.wait_event(ev)
                    return fn(self._side.cuda_stream)
                bw.append(run)
                binfo.append(info)
            defer["ops"] = []

        def join_side():
            ev = torch.cuda.Event()

            def run(s, ev=ev):
                if self._overlap_on:
                    ev.record(self._side)
                    torch.cuda.current_stream().wait_event(ev)
                return 0
            bw.append(run)
            binfo.append(dict(kind="sync", flops=0.0, bytes=0.0))
            for t in defer["puts"]:
                pool.put(t)
            defer["puts"] = []

        # Residual adds folded into the next GroupNorm-backward apply pass (round 4; RHO_FOLD_ADD=0: a pass of their own as before):
        # `G[res] += dY` of a residual connection whose target already holds a gradient is NOT launched; the addend waits here
        # until the apply pass that accumulates into G[res] anyway takes it as `add1` (one read instead of read + read + write).
        # Any other access to G[res] flushes it as the plain rho_add_inplace first.
        fold_add = os.environ.get("RHO_FOLD_ADD", "1") != "0"
        pending_add: Dict[int, Tensor] = {}
        skip_partner: Dict[int, int] = {}               # id(1x1x1 skip node) -> id(the block's in-conv node), see fuse_skip_dgrad
        held_skips: Dict[int, tuple] = {}               # id(in-conv node) -> (skip node, dY, width) waiting for its GroupNorm backward

        def flush_add(k: int):
            src = pending_add.pop(k, None)
            if src is not None:
                a = (ptr(G[k]), ptr(src), dtc, src.numel())
                emit(lambda s, a=a: L.rho_add_inplace(*a, s), "add", nbytes=3.0 * esz * src.numel())
                if src.data_ptr() not in {g_.data_ptr() for g_ in G.values()}:
                    pool.put(src)

        def gradbuf(t: Tensor, fold_ok: bool = False):
            """(buffer, accumulate?) for a write into the gradient of activation t."""
            k = key(t)
            if not fold_ok:
                flush_add(k)
            if k in G:
                return G[k], (k in written)
            G[k] = pool.get(tuple(t.shape), t.dtype)
            return G[k], False

        # scratch shared by all layers (stream-ordered reuse)
        max_w = max(cw.taps * cw.coutp * cw.cinp for cw in eng._convs)
        dwbuf = torch.empty(max_w, dtype=torch.float32, device=dev)
        max_c = max(max(cw.coutp, cw.cinp) for cw in eng._convs)
        nc_tmp = torch.empty(B * max(max_c, 64), dtype=torch.float32, device=dev)
        c_tmp = torch.empty(max(max_c, 64), dtype=torch.float32, device=dev)
        self.dfilm = torch.empty(B, max(eng.film_total, 1), dtype=torch.float32, device=dev)
        self.demb = torch.empty(B, 4 * eng.mc, dtype=torch.float32, device=dev)
        self.demb_h = torch.empty(B, 4 * eng.mc, dtype=torch.float32, device=dev)
        self.keep.extend([dwbuf, nc_tmp, c_tmp])
        film_stride = self.film.shape[1]

        def pgrad(p: nn.Parameter) -> int:
            return p.grad.data_ptr()       # resolved at launch time: optimizers may re-home .grad

        # Deterministic training (rho_set_deterministic / RHO_DETERMINISTIC=1): the weight gradient flushes through ordered slabs
        # (rho_conv_nd_wgrad_ws) instead of fp32 atomics; one workspace per plan, sized for its largest launch after all are known
        self.deterministic = ops.deterministic()
        det_ws: Dict[str, object] = {"bytes": 0, "t": None}

        def wgrad_call(d, dy_ptr: int, w_: int, dw_ptr: Callable[[], int], db_ptr: Callable[[], int]):
            if not self.deterministic:
                return lambda s: L.rho_conv_nd_wgrad(C.byref(d), dy_ptr, w_, dw_ptr(), db_ptr(), s)
            det_ws["bytes"] = max(det_ws["bytes"], int(L.rho_conv_wgrad_workspace_bytes(C.byref(d), w_)))
            return lambda s: L.rho_conv_nd_wgrad_ws(C.byref(d), dy_ptr, w_, dw_ptr(), db_ptr(), ptr(det_ws["t"]), det_ws["bytes"], s)
        self._det_ws = det_ws

        # ---- weight-gradient accumulation ARENA (round 4; RHO_DW_ARENA=0 restores the shared scratch): every weight-gradient launch
        # accumulates into a region of its own ([taps][coutp][cinp] fp32 + the channel sums), the whole arena is cleared by ONE memset
        # at the head of the backward, and the regions are moved into the parameter gradients by ONE table-driven launch per ~32 MiB
        # of parameters (rho_wgrad_finalize_batch) - instead of a memset + finalize + bias-gradient launch per convolution
        # (~250 launches of a few microseconds each per step).  Parameters are reported final (bwd_marks) after their batch.
        use_arena = os.environ.get("RHO_DW_ARENA", "1") != "0"
        arena: Dict[str, object] = {"t": None, "floats": 0}
        # (a batch closes at 32 MiB of parameters, or a sixteenth of the model where that is smaller, so that data-parallel buckets
        #  still become final - and their all-reduce still starts - well inside the backward)
        fin_batch_bytes = min(int(float(os.environ.get("RHO_FIN_BATCH_MB", "32")) * 2 ** 20),
                              max(1, sum(4 * cw.weight.numel() for cw in eng._convs) // 16))
        fin: Dict[str, object] = {"entries": [], "params": [], "bytes": 0}
        self._arena = arena

        def region(nfloats: int) -> int:
            off = arena["floats"]
            arena["floats"] = off + ((int(nfloats) + 63) // 64) * 64
            return off

        def aptr(off: int) -> Callable[[], int]:
            return lambda: arena["t"].data_ptr() + 4 * off

        def fin_add(**e):
            fin["entries"].append(e)

        def fin_close(force: bool = False):
            """Emit the batched finalize of the pending regions (and report their parameters final) once enough bytes are pending."""
            if not fin["entries"] or (not force and fin["bytes"] < fin_batch_bytes):
                return
            entries, params = fin["entries"], fin["params"]
            fin["entries"], fin["params"], fin["bytes"] = [], [], 0
            state = {"sig": None, "dev": None, "blocks": 0}

            def run(s, entries=entries, state=state):
                sig = (arena["t"].data_ptr(),) + tuple(pgrad(e["param"]) for e in entries)
                if sig != state["sig"]:
                    raw, blk = [], 0
                    for e in entries:
                        op = hip.WfinOp()
                        op.dw, op.grad, op.row_src = arena["t"].data_ptr() + 4 * e["off"], pgrad(e["param"]), e["rs"]
                        op.cout, op.cin, op.coutp, op.cinb = e["cout"], e["cin"], e["coutp"], e["cinb"]
                        op.kd, op.kh, op.kw = e["k"]
                        op.total = e["cout"] * e["cin"] * e["k"][0] * e["k"][1] * e["k"][2]
                        op.kind, op.up_h, op.up_w, op.phase_stride = e["kind"], e.get("up_h", 0), e.get("up_w", 0), e.get("stride", 0)
                        op.nblk = max(1, min((op.total + 255) // 256, 512))
                        op.blk0 = blk
                        blk += op.nblk
                        raw.append(bytes(op))
                    state["dev"] = torch.frombuffer(bytearray(b"".join(raw)), dtype=torch.uint8).to(dev)
                    state["blocks"], state["sig"] = blk, sig
                return L.rho_wgrad_finalize_batch(state["dev"].data_ptr(), len(entries), state["blocks"], s)
            emit(run, "wgrad_finalize", nbytes=12.0 * sum(e["cout"] * e["cin"] * e["k"][0] * e["k"][1] * e["k"][2] for e in entries))
            self.keep.append(state)
            self.bwd_marks.append((len(bw), params))

        if use_arena:
            emit(lambda s: (arena["t"].zero_(), 0)[1], "memset")
            arena_memset_info = binfo[-1]

        def bias_and_wgrad(node, dY: Tensor, dyw: int):
            cw = node["cw"]
            N, Do, Ho, Wo = node["out_dims"]
            S = Do * Ho * Wo
            nblk = ops.gn_nblk(S)
            part = pool.get((N * nblk * (dyw // 8) * 16,), torch.float32)
            rs = ptr(cw.row_src)
            # bias gradient = channel sums of dY: accumulated by the weight-gradient kernel itself (below)
            if node["res_add_off"] is not None:        # additive timestep embedding (unet_v2.py:291)
                dst = self.dfilm.data_ptr() + 4 * node["res_add_off"]
                a = (ptr(dY), dtc, N, S, dyw, ptr(part), dst, film_stride, 0, None, 0)
                emit(lambda s, a=a: L.rho_chan_sum(*a, s), "chan_sum", nbytes=float(esz) * N * S * dyw)
            wput(part)
            if node.get("phased") and node["pre"] is None and node["x2"] is None and self.phase_upsample_bwd:
                # Upsample + conv ran as sub-pixel phases: per phase a 2-tap weight gradient on the SOURCE tensor against that parity
                # of dY (12 / 27 of the multiply-adds, no upsampled copy), routed back to the 3-tap parameter gradient
                x1 = node["x1"]
                if use_arena:
                    nwp = max(kk[0] * kk[1] * kk[2] for kk in [(cw.kernel[0], 2 if ph_[0] else cw.kernel[1], 2 if ph_[1] else cw.kernel[2])
                                                                for ph_, _ in cw.wph]) * cw.coutp * cw.cinp
                    stride_ = ((nwp + 63) // 64) * 64
                    off_b, off0 = region(max(dyw, cw.coutp)), region(stride_ * len(cw.wph))
                    for idx, (ph, wt) in enumerate(cw.wph):
                        kern = (cw.kernel[0], 2 if ph[0] else cw.kernel[1], 2 if ph[1] else cw.kernel[2])
                        d = ops.make_conv_desc(x1, None, wt, cw.b, kernel=kern, cout=cw.cout, split=cw.cout, y=dY, y2=None, phase_hw=ph)
                        self.keep.append(d)
                        self.wgrad_descs.append((d, dyw))
                        emit(wgrad_call(d, ptr(dY), dyw, aptr(off0 + idx * stride_), aptr(off_b)), "wgrad",
                             flops=2.0 * N * S * cw.cout * cw.cin * cw.taps / len(cw.wph),
                             nbytes=float(esz) * (x1.numel() + dY.numel() / len(cw.wph)),
                             cin=cw.cin, cout=cw.cout, taps=cw.taps, positions=N * S // len(cw.wph))
                    up_h_, up_w_ = int(any(ph[0] for ph, _ in cw.wph)), int(any(ph[1] for ph, _ in cw.wph))
                    fin_add(kind=1, off=off0, param=cw.weight, rs=None, cout=cw.cout, cin=cw.cin, coutp=cw.coutp, cinb=cw.cinp, k=cw.kernel,
                            up_h=up_h_, up_w=up_w_, stride=stride_)
                    fin_add(kind=0, off=off_b, param=cw.bias_param, rs=rs, cout=cw.cout, cin=1, coutp=dyw, cinb=1, k=(1, 1, 1))
                    fin["params"] += [cw.weight, cw.bias_param]
                    fin["bytes"] += 4 * cw.weight.numel()
                    return
                cbv = c_tmp[:max(dyw, cw.coutp)]
                emit(lambda s, t2=cbv: (t2.zero_(), 0)[1], "memset", nbytes=4.0 * cbv.numel())
                for ph, wt in cw.wph:
                    kern = (cw.kernel[0], 2 if ph[0] else cw.kernel[1], 2 if ph[1] else cw.kernel[2])
                    d = ops.make_conv_desc(x1, None, wt, cw.b, kernel=kern, cout=cw.cout, split=cw.cout, y=dY, y2=None, phase_hw=ph)
                    self.keep.append(d)
                    self.wgrad_descs.append((d, dyw))
                    nwp = kern[0] * kern[1] * kern[2] * cw.coutp * cw.cinp
                    dwv = dwbuf[:nwp]
                    emit(lambda s, t=dwv: (t.zero_(), 0)[1], "memset", nbytes=4.0 * nwp)
                    emit(wgrad_call(d, ptr(dY), dyw, lambda: ptr(dwbuf), lambda: ptr(c_tmp)), "wgrad",
                         flops=2.0 * N * S * cw.cout * cw.cin * cw.taps / len(cw.wph),
                         nbytes=float(esz) * (x1.numel() + dY.numel() / len(cw.wph)),
                         cin=cw.cin, cout=cw.cout, taps=cw.taps, positions=N * S // len(cw.wph))
                    emit(lambda s, cw=cw, ph=ph: L.rho_wgrad_finalize_phase(ptr(dwbuf), pgrad(cw.weight), cw.cout, cw.cin, cw.kernel[0],
                                                                           cw.kernel[1], cw.kernel[2], ph[0], ph[1], cw.coutp, cw.cinp, 1, s),
                         "wgrad_finalize", nbytes=8.0 * nwp)
                emit(lambda s, cw=cw, rs=rs, w_=dyw: L.rho_wgrad_finalize(ptr(c_tmp), pgrad(cw.bias_param), cw.cout, 1, 1, w_, 1, rs, 1, s),
                     "bias_grad")
                return
            # weight gradient (forward descriptor; upsampled input materialised)
            x1 = node["x1"]
            tmp_up = None
            if node["up_hw"] != (0, 0):
                uh, uw = node["up_hw"]
                tmp_up = pool.get((x1.shape[0], x1.shape[1], x1.shape[2] * (2 if uh else 1), x1.shape[3] * (2 if uw else 1),
                                   x1.shape[4]), dt)
                a = (ptr(x1), ptr(tmp_up), dtc, x1.shape[0] * x1.shape[1], x1.shape[2], x1.shape[3], x1.shape[4], int(uh), int(uw))
                emit(lambda s, a=a: L.rho_upsample2x(*a, s), "upsample", nbytes=5.0 * esz * x1.numel())
                x1 = tmp_up
            pre = node["pre"]
            x2 = node["x2"]
            xact = None
            if node.get("xact") is not None:
                x1, x2 = node["xact"], None              # materialised by the forward plan
            elif pre is not None:
                # materialise act(a*x+b) once (HBM-rate) instead of redoing it in every (cout tile, cin chunk) workgroup
                c1_ = x1.shape[-1]
                c2_ = x2.shape[-1] if x2 is not None else 0
                xact = pool.get(tuple(x1.shape[:4]) + (c1_ + c2_,), dt)
                Nn = x1.shape[0]
                Sx = x1.shape[1] * x1.shape[2] * x1.shape[3]
                a = (ptr(x1), c1_, ptr(x2), c2_, dtc, Nn, Sx, ptr(pre["a"]), ptr(pre["b"]), int(node["pre_silu"]), ptr(xact))
                if node.get("drop") is not None:     # the same mask as the forward: same key, same counter
                    ad = a + (float(node["drop"][0]), int(node["drop"][1]), ptr(self.drop_ctr))
                    emit(lambda s, a=ad: L.rho_gn_apply_drop(*a, s), "gn_apply", nbytes=2.0 * esz * xact.numel())
                else:
                    emit(lambda s, a=a: L.rho_gn_apply(*a, s), "gn_apply", nbytes=2.0 * esz * xact.numel())
                x1, x2 = xact, None
            d = ops.make_conv_desc(x1, x2, cw.w, cw.b, kernel=cw.kernel, cout=cw.cout, split=cw.cout, y=dY, y2=None,
                                   stride_hw=node["stride_hw"], pre_silu=False)
            self.keep.append(d)
            self.wgrad_descs.append((d, dyw))
            nw = cw.taps * cw.coutp * cw.cinp
            if use_arena:
                off_w, off_b = region(nw), region(max(dyw, cw.coutp))
                emit(wgrad_call(d, ptr(dY), dyw, aptr(off_w), aptr(off_b)), "wgrad",
                     flops=2.0 * N * S * cw.cout * cw.cin * cw.taps, nbytes=float(esz) * (x1.numel() + dY.numel()),
                     cin=cw.cin, cout=cw.cout, taps=cw.taps, positions=N * S)
                fin_add(kind=0, off=off_w, param=cw.weight, rs=rs, cout=cw.cout, cin=cw.cin, coutp=cw.coutp, cinb=cw.cinp, k=cw.kernel)
                fin_add(kind=0, off=off_b, param=cw.bias_param, rs=rs, cout=cw.cout, cin=1, coutp=dyw, cinb=1, k=(1, 1, 1))
                fin["params"] += [cw.weight, cw.bias_param]
                fin["bytes"] += 4 * cw.weight.numel()
                if tmp_up is not None:
                    wput(tmp_up)
                if xact is not None:
                    wput(xact)
                return
            dwv = dwbuf[:nw]
            cbv = c_tmp[:max(dyw, cw.coutp)]
            emit(lambda s, t=dwv, t2=cbv: (t.zero_(), t2.zero_(), 0)[2], "memset", nbytes=4.0 * nw)
            emit(wgrad_call(d, ptr(dY), dyw, lambda: ptr(dwbuf), lambda: ptr(c_tmp)), "wgrad",
                 flops=2.0 * N * S * cw.cout * cw.cin * cw.taps, nbytes=float(esz) * (x1.numel() + dY.numel()),
                 cin=cw.cin, cout=cw.cout, taps=cw.taps, positions=N * S)
            emit(lambda s, cw=cw, rs=rs: L.rho_wgrad_finalize(ptr(dwbuf), pgrad(cw.weight), cw.cout, cw.cin, cw.taps, cw.coutp,
                                                              cw.cinp, rs, 1, s), "wgrad_finalize", nbytes=8.0 * nw)
            # (row-permuted, truncated) accumulate of the channel sums into bias.grad
            emit(lambda s, cw=cw, rs=rs, w_=dyw: L.rho_wgrad_finalize(ptr(c_tmp), pgrad(cw.bias_param), cw.cout, 1, 1, w_, 1, rs, 1, s),
                 "bias_grad")
            if tmp_up is not None:
                wput(tmp_up)
            if xact is not None:
                wput(xact)

        def gn_backward(pre, pre_silu, x1, x2, dact, fused=None, drop=None, skip=None):
            """dact = gradient of act(GroupNorm(x) * (1 + scale) + shift): reduce / finalize / apply into the gradients of x1 (, x2),
            the norm's parameters and the FiLM rows.  ``fused`` = (tile sums, tiles per sample) when the dgrad launch that produced
            dact already reduced dz and dz * x in its epilogue (rho_conv_desc.gnb_*): the reduce pass is skipped.  ``skip`` = (node,
            dY, width) of the block's 1x1x1 skip convolution whose data gradient was held back: it runs here, with the apply pass in
            its epilogue (rho_conv_desc.gna_*), or - where that launch does not exist - on its own in front of the apply pass."""
            c1 = x1.shape[-1]
            c2 = x2.shape[-1] if x2 is not None else 0
            norm = pre["norm"]
            Cc, N_, S_ = pre["C"], pre["N"], pre["S"]
            skip_fused = None
            if skip is not None:
                sk_node, sk_dY, sk_w = skip
                can = (drop is None and int(pre_silu) <= 1 and key(x1) not in written and key(x1) not in pending_add
                       and (x2 is None or key(x2) not in written) and S_ % 256 == 0 and Cc % 32 == 0
                       and (x2 is None or c2 % (8 if dt == torch.bfloat16 else 4) == 0))
                if can:
                    skip_fused = skip
                else:
                    dgrad(sk_node, sk_dY, sk_w, hold_skip=False)          # the two-pass form: data gradient first, apply accumulates
                    pool.put(sk_dY)
            g1, acc1 = gradbuf(x1, fold_ok=True)
            add1 = pending_add.pop(key(x1), None)              # a residual's gradient waiting to join G[x1]: folded into this pass
            g2, acc2 = gradbuf(x2) if x2 is not None else (None, False)
            cA = pool.get((N_, Cc), torch.float32)
            cP = pool.get((N_, 32), torch.float32)
            cQ = pool.get((N_, 32), torch.float32)
            work = pool.get((2 * N_ * Cc,), torch.float32)
            scale = dscale = dshift = None
            fstride = 0
            if pre["film_off"] is not None:
                scale = self.film.data_ptr() + 4 * pre["film_off"]
                fstride = film_stride
                dscale = self.dfilm.data_ptr() + 4 * pre["film_off"]
                dshift = self.dfilm.data_ptr() + 4 * (pre["film_off"] + Cc)
            if fused is None:
                a1 = (ptr(dact), ptr(x1), c1, ptr(x2), c2, dtc, N_, S_, ptr(pre["a"]), ptr(pre["b"]), ptr(pre["st"]),
                      int(pre_silu), ptr(pre["part"]))
                if drop is not None:
                    a1d = a1 + (float(drop[0]), int(drop[1]), ptr(self.drop_ctr))
                    emit(lambda s, a=a1d: L.rho_gn_bwd_reduce_drop(*a, s), "gn_bwd_reduce", nbytes=2.0 * esz * N_ * S_ * Cc)
                else:
                    emit(lambda s, a=a1: L.rho_gn_bwd_reduce(*a, s), "gn_bwd_reduce", nbytes=2.0 * esz * N_ * S_ * Cc)
                part_ptr, part_n, fmt = ptr(pre["part"]), pre["nblk"], 0
            else:
                part_ptr, part_n, fmt = ptr(fused[0]), fused[1], 1
            emit(lambda s, pre=pre, norm=norm, scale=scale, fstride=fstride, work=work, dscale=dscale, dshift=dshift,
                 cA=cA, cP=cP, cQ=cQ, N_=N_, Cc=Cc, S_=S_, part_ptr=part_ptr, part_n=part_n, fmt=fmt: L.rho_gn_bwd_finalize(
                     part_ptr, N_, Cc, S_, part_n, fmt, ptr(norm.weight), ptr(norm.bias), scale, fstride, ptr(pre["st"]),
                     ptr(work), pgrad(norm.weight), pgrad(norm.bias), 1, dscale, dshift, film_stride, ptr(cA), ptr(cP),
                     ptr(cQ), s), "gn_bwd_finalize")
            a3 = (ptr(dact), ptr(x1), c1, ptr(x2), c2, dtc, N_, S_, ptr(pre["a"]), ptr(pre["b"]), int(pre_silu),
                  ptr(cA), ptr(cP), ptr(cQ), ptr(g1), ptr(g2), int(acc1), int(acc2), ptr(add1))
            nb3 = esz * N_ * S_ * (3.0 * Cc + (c1 if acc1 else 0) + (c2 if acc2 else 0) + (c1 if add1 is not None else 0))
            if skip_fused is not None:
                # dX = skip^T(dY) + [cA * (dact * act'(a x + b)) + cQ * x + cP] in the 1x1x1 data-gradient launch's epilogue
                sk_node, sk_dY, sk_w = skip_fused
                scw = sk_node["cw"]
                if acc1 or acc2 or add1 is not None:
                    raise hip.RhoHipError("internal: fused skip data gradient on a gradient that already has a writer (backward plan)")
                d = ops.make_conv_desc(sk_dY, None, scw.wd, scw.zero_bias, kernel=scw.kernel, cout=Cc, split=c1, y=g1, y2=g2,
                                       y2_cl=x2 is not None)
                d.gnb_x1, d.gnb_x2, d.gnb_c1, d.gnb_silu = ptr(x1), ptr(x2), c1, int(pre_silu)
                d.gnb_a, d.gnb_b = ptr(pre["a"]), ptr(pre["b"])
                d.gna_g, d.gna_cA, d.gna_cP, d.gna_cQ = ptr(dact), ptr(cA), ptr(cP), ptr(cQ)
                self.keep.append(d)
                self.fwd_descs.append(d)
                emit(lambda s, d=d: L.rho_conv_nd_fwd(C.byref(d), s), "dgrad", flops=2.0 * N_ * S_ * Cc * scw.cout,
                     nbytes=float(esz) * (sk_dY.numel() + 3.0 * N_ * S_ * Cc))
                pool.put(sk_dY)
            elif drop is not None:
                a3d = a3 + (float(drop[0]), int(drop[1]), ptr(self.drop_ctr))
                emit(lambda s, a=a3d: L.rho_gn_bwd_apply_drop(*a, s), "gn_bwd_apply", nbytes=nb3)
            else:
                emit(lambda s, a=a3: L.rho_gn_bwd_apply(*a, s), "gn_bwd_apply", nbytes=nb3)
            if add1 is not None and add1.data_ptr() not in {g_.data_ptr() for g_ in G.values()}:
                pool.put(add1)                                 # (stream order: recycled buffers are written by later launches only)
            written.add(key(x1))
            if x2 is not None:
                written.add(key(x2))
            for t in (cA, cP, cQ, work):
                pool.put(t)

        def dgrad(node, dY: Tensor, dyw: int, after_launch: Optional[Callable[[], None]] = None, hold_skip: bool = True):
            """Data gradient of a conv node.  Returns True when the launch was held back: the 1x1x1 skip convolution of a ResBlock
            whose in-conv path ends in a GroupNorm backward of the same inputs - it runs inside that pass (gn_backward's ``skip``)."""
            cw = node["cw"]
            if hold_skip and id(node) in skip_partner:
                held_skips[skip_partner[id(node)]] = (node, dY, dyw)
                return True
            x1, x2, pre = node["x1"], node["x2"], node["pre"]
            c1 = x1.shape[-1]
            c2 = x2.shape[-1] if x2 is not None else 0
            cin = c1 + c2
            if dyw != cw.wd.shape[2] or cw.wd.shape[1] != cin:
                raise hip.RhoHipError("internal: dgrad weight shape does not match the gradient tensors")
            common = dict(kernel=cw.kernel, cout=cin)
            if node.get("phased") and pre is None and x2 is None and self.phase_upsample_bwd:
                # Upsample + conv ran as sub-pixel phases: each phase's share of dX is a 2-tap conv of that parity of dY with the
                # phase's flipped weights, accumulated in place - 12 / 27 of the multiply-adds, no full-resolution intermediate
                g1, acc1 = gradbuf(x1)
                for i, (ph, wt) in enumerate(cw.wphd):
                    kern = (cw.kernel[0], 2 if ph[0] else cw.kernel[1], 2 if ph[1] else cw.kernel[2])
                    d = ops.make_conv_desc(dY, None, wt, cw.zero_bias, kernel=kern, cout=cin, split=cin, y=g1, y2=None,
                                           res=g1 if (acc1 or i > 0) else None, phase_dgrad_hw=ph)
                    self.keep.append(d)
                    self.fwd_descs.append(d)
                    emit(lambda s, d=d: L.rho_conv_nd_fwd(C.byref(d), s), "dgrad",
                         flops=2.0 * (dY.numel() // dyw) * cin * cw.cout * cw.taps / len(cw.wphd),
                         nbytes=float(esz) * (dY.numel() / len(cw.wphd) + x1.numel() * (2 if (acc1 or i > 0) else 1)))
                written.add(key(x1))
            elif pre is not None or node["up_hw"] != (0, 0):
                N, Do, Ho, Wo = node["out_dims"]
                tshape = (N, Do, Ho, Wo, cin) if node["up_hw"] != (0, 0) else tuple(x1.shape[:4]) + (cin,)
                dact = pool.get(tshape, dt)       # gradient of the activated / upsampled tensor
                d = ops.make_conv_desc(dY, None, cw.wd, cw.zero_bias, split=cin, y=dact, y2=None, **common)
                fused = None
                if (pre is not None and self.fuse_gn_bwd > 0 and cin >= self.fuse_gn_bwd and int(node["pre_silu"]) <= 1
                        and node.get("drop") is None):
                    # the norm's backward reductions ride in this launch's epilogue where a tile lies in one sample
                    tiles = int(L.rho_conv_stats_tiles(C.byref(d)))
                    if tiles > 0:
                        sbuf = pool.get((x1.shape[0] * tiles * 2 * cin,), torch.float32)
                        d.stats = sbuf.data_ptr()
                        d.gnb_x1, d.gnb_x2, d.gnb_c1 = ptr(x1), ptr(x2), c1
                        d.gnb_a, d.gnb_b, d.gnb_silu = ptr(pre["a"]), ptr(pre["b"]), int(node["pre_silu"])
                        fused = (sbuf, tiles)
                self.keep.append(d)
                self.fwd_descs.append(d)
                emit(lambda s, d=d: L.rho_conv_nd_fwd(C.byref(d), s), "dgrad",
                     flops=2.0 * (dact.numel() // cin) * cin * cw.cout * cw.taps, nbytes=float(esz) * (dY.numel() + dact.numel()))
                if after_launch is not None:
                    after_launch()                    # (the deferred weight gradient starts here, on the side stream)
                if pre is not None:
                    gn_backward(pre, node["pre_silu"], x1, x2, dact, fused, drop=node.get("drop"), skip=held_skips.pop(id(node), None))
                    if fused is not None:
                        pool.put(fused[0])
                else:   # upsample: sum the 2x2 (1x2) children
                    g1, acc1 = gradbuf(x1)
                    a = (ptr(dact), ptr(g1), dtc, x1.shape[0] * x1.shape[1], x1.shape[2], x1.shape[3], x1.shape[4],
                         int(node["up_hw"][0]), int(node["up_hw"][1]), int(acc1))
                    emit(lambda s, a=a: L.rho_pool2x_sum(*a, s), "pool2x", nbytes=5.0 * esz * x1.numel())
                    written.add(key(x1))
                pool.put(dact)
            elif node.get("s2") and self.s2_split_bwd:
                # stride-2 conv: one launch per parity of dX (dx[2m] = w1 dy[m]; dx[2m+1] = w2 dy[m] + w0 dy[m+1]) instead of a 27-tap
                # conv over a zero-stuffed dY (three of four multiply-adds on zeros)
                g1, acc1 = gradbuf(x1)
                for (a, b), wt in cw.ws2d:
                    kern = (3, len(cw.S2_BWD[a]), len(cw.S2_BWD[b]))
                    d = ops.make_conv_desc(dY, None, wt, cw.zero_bias, kernel=kern, cout=cin, split=cin, y=g1, y2=None,
                                           res=g1 if acc1 else None, phase_hw=(a + 1, b + 1))
                    self.keep.append(d)
                    self.fwd_descs.append(d)
                    emit(lambda s, d=d: L.rho_conv_nd_fwd(C.byref(d), s), "dgrad",
                         flops=2.0 * (dY.numel() // dyw) * cin * cw.cout * kern[0] * kern[1] * kern[2],
                         nbytes=float(esz) * (dY.numel() + x1.numel() / 4 * (2 if acc1 else 1)))
                written.add(key(x1))
            else:
                g1, acc1 = gradbuf(x1)
                g2, acc2 = gradbuf(x2) if x2 is not None else (None, False)
                st = node["stride_hw"]
                zs = (int(st[0] == 2), int(st[1] == 2))
                d = ops.make_conv_desc(dY, None, cw.wd, cw.zero_bias, split=c1, y=g1, y2=g2, y2_cl=x2 is not None,
                                       res=g1 if acc1 else None, res2=g2 if (x2 is not None and acc2) else None,
                                       zs_hw=zs, out_hw=(x1.shape[2], x1.shape[3]) if zs != (0, 0) else (0, 0), **common)
                self.keep.append(d)
                self.fwd_descs.append(d)
                emit(lambda s, d=d: L.rho_conv_nd_fwd(C.byref(d), s), "dgrad",
                     flops=2.0 * (x1.numel() // c1) * cin * cw.cout * cw.taps, nbytes=float(esz) * (dY.numel() + x1.numel()))
                written.add(key(x1))
                if x2 is not None:
                    written.add(key(x2))

        # ---- head: dpred [N, Cout, S] float32 -> channels-last, as wide as the dgrad weights expect
        self.dpred_in = torch.empty(tuple(self.out.shape), dtype=torch.float32, device=dev)
        head = self.nodes[-1]

        def im2col_of(src_f32: Tensor, dims4):
            """[N, 1, D, H, W] float32 -> channels-last [N, D, H, W, 32] (27 taps + zero pad), the K = 32 operand of the GEMM forms."""
            N_, D_, H_, W_ = dims4
            im = pool.get((N_, D_, H_, W_, 32), dt)
            a_ = (ptr(src_f32), ptr(im), dtc, N_, 1, D_, H_, W_, 3, 3, 3, 32)
            emit(lambda s, a=a_: L.rho_im2col_taps(*a, s), "pack", nbytes=4.0 * src_f32.numel() + float(esz) * im.numel())
            return im

        def gemm_wgrad(x_t: Tensor, dy_t: Tensor, rows: int):
            """dw[rows][cin] (+ channel sums of dy_t) = sum_pos dy_t[pos][:rows] x_t[pos][:] on the GEMM-shaped k_wgrad1; returns the
            arena offsets (weights, channel sums)."""
            cin_ = x_t.shape[-1]
            dummy_w = torch.empty(1, rows, cin_, dtype=dt, device=dev)
            dummy_b = torch.zeros(rows, dtype=torch.float32, device=dev)
            self.keep.extend([dummy_w, dummy_b])
            d = ops.make_conv_desc(x_t, None, dummy_w, dummy_b, kernel=(1, 1, 1), cout=rows, split=rows, y=dy_t, y2=None)
            self.keep.append(d)
            self.wgrad_descs.append((d, dy_t.shape[-1]))
            off_w, off_b = region(rows * cin_), region(max(rows, dy_t.shape[-1]))
            npos_ = x_t.numel() // cin_
            emit(wgrad_call(d, ptr(dy_t), dy_t.shape[-1], aptr(off_w), aptr(off_b)), "wgrad", flops=2.0 * npos_ * 27 * max(rows, cin_),
                 nbytes=float(esz) * (x_t.numel() + dy_t.numel()), cin=cin_, cout=rows, taps=1, positions=npos_)
            return off_w, off_b

        if head["k"] == "head_direct":
            # one-output-channel head conv (unet_v2.py:679-683): data gradient = rho_stem_conv3d on dpred with mirrored taps; weight
            # gradient = GEMM of the kept activated input against the im2col of dpred (rows = taps, mirrored back by finalize kind 2);
            # bias gradient = the im2col's centre column sum (tap 13 never touches the padding) = sum of dpred
            hcw = head["cw"]
            N, Do, Ho, Wo = head["out_dims"]
            Ch = head["x"].shape[-1]
            hdw = eng._head_dgrad(m.out[2])
            dact = pool.get((N, Do, Ho, Wo, Ch), dt)
            a = (ptr(self.dpred_in), ptr(hdw.w), ptr(hdw.zero_bias), ptr(dact), None, N, Do, Ho, Wo, Ch)
            emit(lambda s, a=a: L.rho_stem_conv3d(*a, s), "dgrad", flops=2.0 * N * Do * Ho * Wo * Ch * 27,
                 nbytes=4.0 * N * Do * Ho * Wo + float(esz) * dact.numel())
            im = im2col_of(self.dpred_in, (N, Do, Ho, Wo))
            off_w, off_b = gemm_wgrad(head["xact"], im, 32)
            pool.put(im)
            fin_add(kind=2, off=off_w, param=hcw.weight, rs=None, cout=1, cin=Ch, coutp=32, cinb=Ch, k=(3, 3, 3))   # (walks [ci][tap])
            fin_add(kind=0, off=off_b + 13, param=hcw.bias_param, rs=None, cout=1, cin=1, coutp=1, cinb=1, k=(1, 1, 1))
            fin["params"] += [hcw.weight, hcw.bias_param]
            fin["bytes"] += 4 * hcw.weight.numel()
            gn_backward(head["pre"], eng.act, head["x"], None, dact)
            pool.put(dact)
            self.bwd_marks.append((len(bw), [head["pre"]["norm"].weight, head["pre"]["norm"].bias]))
        else:
            hw = head["cw"].wd.shape[2]
            N, Do, Ho, Wo = head["out_dims"]
            dhead = pool.get((N, Do, Ho, Wo, hw), dt)
            a = (ptr(self.dpred_in), ptr(dhead), dtc, N, head["cw"].cout, Do * Ho * Wo, hw)
            emit(lambda s, a=a: L.rho_pack_input(*a, s), "pack")
            G[key(head["y2"])] = dhead
            written.add(key(head["y2"]))

        rs_hw = (1, 1) if eng.dims >= 2 else (0, 1)
        # 1x1x1 skip-conv nodes whose block input also feeds a normalised conv (the block's in-conv, an earlier node): id(skip node) ->
        # id(in-conv node).  Their data gradient is held until that node's GroupNorm backward (fuse_skip_dgrad)
        if self.fuse_skip_dgrad:
            for i, nd in enumerate(self.nodes):
                if (nd["k"] == "conv" and nd["cw"].taps == 1 and nd["pre"] is None and not nd["stem"] and nd["up_hw"] == (0, 0)
                        and tuple(nd["stride_hw"]) == (1, 1) and nd["res"] is None and not nd.get("phased") and not nd.get("s2")):
                    for pj in range(i - 1, -1, -1):
                        pn = self.nodes[pj]
                        if (pn["k"] == "conv" and pn["pre"] is not None and pn["x1"] is nd["x1"] and pn["x2"] is nd["x2"]
                                and pn["up_hw"] == (0, 0) and not pn["stem"]):
                            skip_partner[id(nd)] = id(pn)
                            break
        for node in reversed(self.nodes):
            if node["k"] == "head_direct":
                continue                                       # (handled above)
            if node["k"] == "stem_direct":
                # one-input-channel stem conv (unet_v2.py:535): weight gradient = GEMM of the im2col of the input against dY
                # (dw[co][tap], the parameter's own layout), bias gradient = channel sums of dY; no data gradient
                flush_add(key(node["y"]))
                dY = G.pop(key(node["y"]), None)
                if dY is None:
                    raise hip.RhoHipError("internal: missing gradient of the stem output in the backward plan")
                scw = node["cw"]
                im = im2col_of(self.x_in, node["out_dims"])
                off_w, off_b = gemm_wgrad(im, dY, scw.cout)
                pool.put(im)
                pool.put(dY)
                fin_add(kind=0, off=off_w, param=scw.weight, rs=None, cout=scw.cout, cin=27, coutp=scw.cout, cinb=32, k=(1, 1, 1))
                fin_add(kind=0, off=off_b, param=scw.bias_param, rs=None, cout=scw.cout, cin=1, coutp=dY.shape[-1], cinb=1, k=(1, 1, 1))
                fin["params"] += [scw.weight, scw.bias_param]
                fin["bytes"] += 4 * scw.weight.numel()
                fin_close()
                continue
            if node["k"] == "resample":
                # y = avgpool / nearest-upsample(x): the transposed map into the gradient of x (accumulating if x has other consumers)
                flush_add(key(node["y"]))
                dy = G.pop(key(node["y"]), None)
                if dy is None:
                    raise hip.RhoHipError("internal: missing gradient of a resampled tensor in the backward plan")
                xt = node["x"]
                gx, accx = gradbuf(xt)
                a = (ptr(dy), ptr(gx), dtc, xt.shape[0] * xt.shape[1], xt.shape[2], xt.shape[3], xt.shape[4], rs_hw[0], rs_hw[1], int(accx))
                if node["mode"] == "up":
                    emit(lambda s, a=a: L.rho_pool2x_sum(*a, s), "pool2x", nbytes=5.0 * esz * xt.numel())
                else:
                    emit(lambda s, a=a: L.rho_avgpool2x_bwd(*a, s), "avgpool_bwd", nbytes=3.0 * esz * xt.numel())
                written.add(key(xt))
                pool.put(dy)
                continue
            if node["k"] == "act":
                flush_add(key(node["y"]))
                dact = G.pop(key(node["y"]), None)
                if dact is None:
                    raise hip.RhoHipError("internal: missing gradient of a materialised activation in the backward plan")
                gn_backward(node["pre"], node["pre_silu"], node["x1"], node["x2"], dact)
                pool.put(dact)
                pre = node["pre"]
                self.bwd_marks.append((len(bw), [pre["norm"].weight, pre["norm"].bias]))
                continue
            if node["k"] == "attn":
                flush_add(key(node["ao"]))
                dao = G.get(key(node["ao"]))
                N, T, Cc = node["N"], node["T"], node["C"]
                dqkv = pool.get(tuple(node["qk"].shape[:4]) + (3 * Cc,), dt)   # = dY of the qkv projection
                delta = pool.get((N, node["heads"], T), torch.float32)
                a = (ptr(node["qk"]), ptr(node["vt"]), ptr(node["ao"]), ptr(dao), ptr(node["lse"]), ptr(delta), dqkv.data_ptr(), 3 * Cc,
                     dqkv.data_ptr() + 2 * Cc * esz, 3 * Cc, dtc, N, T, node["heads"], Cc // node["heads"])
                emit(lambda s, a=a: L.rho_attention_bwd(*a, s), "attention_bwd", flops=14.0 * N * T * T * Cc)
                G[key(node["qk"])] = dqkv
                written.add(key(node["qk"]))
                pool.put(delta)
                pool.put(G.pop(key(node["ao"])))
                continue
            cw = node["cw"]
            out_t = node["y"] if node["y"] is not None else node["y2"]
            flush_add(key(out_t))
            dY = G.get(key(out_t))
            if dY is None:
                raise hip.RhoHipError("internal: missing output gradient in backward plan")
            dyw = dY.shape[-1]
            held_for_add = False
            # residual input of the epilogue: alias (first contribution) or accumulate
            if node["res"] is not None:
                rk = key(node["res"])
                if rk not in G:
                    G[rk] = dY
                    written.add(rk)
                elif fold_add and rk not in pending_add and G[rk].shape == dY.shape:
                    pending_add[rk] = dY                       # joins G[rk] in the next apply pass that accumulates into it
                    held_for_add = True
                else:
                    flush_add(rk)
                    a = (ptr(G[rk]), ptr(dY), dtc, dY.numel())
                    emit(lambda s, a=a: L.rho_add_inplace(*a, s), "add", nbytes=3.0 * esz * dY.numel())
            # (normalised convs: the weight gradient runs beside the GroupNorm backward passes, see `overlap` above)
            ov = overlap and node["pre"] is not None and not node["stem"] and node["up_hw"] == (0, 0)
            defer["on"] = ov
            bias_and_wgrad(node, dY, dyw)
            defer["on"] = False
            held_skip = False
            if not node["stem"]:
                held_skip = bool(dgrad(node, dY, dyw, after_launch=flush_side if ov else None))
            if ov:
                if defer["ops"]:
                    raise hip.RhoHipError("internal: deferred weight-gradient launches were never issued (backward plan)")
                join_side()
            # the output gradient is dead now unless a residual aliased it
            aliased = node["res"] is not None and G.get(key(node["res"])) is dY
            G.pop(key(out_t), None)
            if not aliased and not held_for_add and not held_skip:
                pool.put(dY)
            ps = [] if use_arena else [cw.weight, cw.bias_param]      # (arena: reported with their finalize batch)
            if node["pre"] is not None:
                ps += [node["pre"]["norm"].weight, node["pre"]["norm"].bias]
            if ps:
                self.bwd_marks.append((len(bw), ps))
            if use_arena:
                fin_close()

        if pending_add:
            raise hip.RhoHipError("internal: a residual gradient was never added (backward plan)")
        if held_skips:
            raise hip.RhoHipError("internal: a held skip data gradient was never launched (backward plan)")
        if use_arena:
            fin_close(force=True)
            arena["t"] = torch.empty(max(arena["floats"], 64), dtype=torch.float32, device=dev)
            arena_memset_info["bytes"] = 4.0 * arena["t"].numel()
            self.arena_bytes = 4 * arena["t"].numel()           # (a fixed cost of the model's size, whatever the plan keeps of activations)
            pool.all.append(arena["t"])
        # ---- embedding path (needs the FiLM gradients of every block)
        e = 4 * eng.mc
        first = True
        emb_params: List[nn.Parameter] = []
        for blk in eng._film_blocks:
            lin = blk.emb_layers[1]
            off = eng._film_off[id(blk)]
            O = lin.weight.shape[0]
            dptr = self.dfilm.data_ptr() + 4 * off
            emit(lambda s, lin=lin, dptr=dptr, O=O, first=first: L.rho_linear_bwd(
                dptr, film_stride, ptr(self.emb), ptr(lin.weight), pgrad(lin.weight), pgrad(lin.bias), ptr(self.demb), B, e, O, eng.act, 1,
                0 if first else 1, s), "linear_bwd", flops=4.0 * B * e * O)
            first = False
            emb_params += [lin.weight, lin.bias]
        te0, te2 = m.time_embed[0], m.time_embed[2]
        emit(lambda s: L.rho_linear_bwd(ptr(self.demb), 0, ptr(self.emb_h), ptr(te2.weight), pgrad(te2.weight), pgrad(te2.bias),
                                        ptr(self.demb_h), B, e, e, eng.act, 1, 0, s), "linear_bwd")
        emit(lambda s: L.rho_linear_bwd(ptr(self.demb_h), 0, ptr(self.sin_in), ptr(te0.weight), pgrad(te0.weight), pgrad(te0.bias),
                                        None, B, eng.mc, e, 0, 1, 0, s), "linear_bwd")
        emb_params += [te2.weight, te2.bias, te0.weight, te0.bias]
        self.bwd_marks.append((len(bw), emb_params))
        if self.deterministic and det_ws["bytes"] > 0:
            det_ws["t"] = torch.empty((det_ws["bytes"] + 3) // 4, dtype=torch.float32, device=dev)
            pool.all.append(det_ws["t"])
        self.pool_bytes = sum(t.numel() * t.element_size() for t in pool.all)

    def nbytes(self) -> int:
        """Device bytes this plan owns (forward buffers kept for its lifetime + the backward pool)."""
        seen, tot = set(), 0
        for t in self.keep:
            if torch.is_tensor(t) and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                tot += t.numel() * t.element_size()
        return tot + int(getattr(self, "pool_bytes", 0))

    def variants(self) -> List[str]:
        """Names of the k_conv / k_wgrad instantiations this plan launches (rho_conv_variant / rho_conv_wgrad_variant)."""
        out = [ops.conv_variant(d) for d in self.fwd_descs]
        out += [ops.conv_wgrad_variant(d, w_) for d, w_ in self.wgrad_descs]
        return out

    # ------------------------------------------------------------------ execution
    def run(self, x: Tensor, timesteps: Optional[Tensor], y: Optional[Tensor], t_scalar_dev: Optional[Tensor]) -> Tensor:
        eng = self.eng
        m = eng.model
        if x.data_ptr() != self.x_in.data_ptr():
            self.x_in.copy_(x)
        te0, te2 = m.time_embed[0], m.time_embed[2]
        self.cond_src = None
        self.cond_dev = None
        if self.cond is not None:
            # label handling of unet_v2.py:702-719
            if y.dim() == 2 and tuple(y.shape) == tuple(self.emb.shape):
                self.cond.copy_(y.to(self.cond.device))
            else:
                if y.dim() == 1:
                    assert y.shape == (x.shape[0],)
                else:
                    assert y.shape[0] == self.emb.shape[0]
                cd = eng.cond_device_tables()
                if cd is not None:
                    # MultiEmbeddings on the device: category lookup by exact equality + row sum, no host synchronisation
                    yf = y.to(device=self.cond.device, dtype=torch.float32).contiguous()
                    if yf.dim() == 2 and yf.shape[1] != cd["nkeys"]:
                        raise hip.RhoHipError(f"labels have {yf.shape[1]} columns, the parameter space has {cd['nkeys']} keys")
                    self.keep_y = yf
                    check(self.L.rho_multi_embed(ptr(yf), 1 if yf.dim() == 1 else yf.shape[1], ptr(cd["space"]), ptr(cd["key_off"]),
                                                 ptr(cd["tables"]), cd["nkeys"], self.B, self.cond.shape[1], ptr(self.cond),
                                                 ptr(self.cond_idx), ptr(eng.err_flag()), hip.stream()), "rho_multi_embed")
                    self.cond_dev = cd
                else:
                    with (torch.enable_grad() if self.train else torch.no_grad()):
                        c = m.cond_fn(y)          # a user-supplied cond_fn module (not MultiEmbeddings): evaluated as given
                    self.cond_src = c if (self.train and c.requires_grad) else None
                    self.cond.copy_(c.detach())
        # timestep embedding: sinusoid of t (any integer) -> Linear -> SiLU -> Linear (+ cond), one launch
        if t_scalar_dev is not None:
            tptr, tsptr = None, ptr(t_scalar_dev)
        else:
            hip.require_gpu(timesteps, "timesteps")
            self.t_in.copy_(timesteps.reshape(-1))
            tptr, tsptr = ptr(self.t_in), None
        check(self.L.rho_timestep_embed(ptr(eng.omega()), tptr, tsptr, ptr(te0.weight), ptr(te0.bias), ptr(te2.weight), ptr(te2.bias),
                                        ptr(self.cond), ptr(self.sin_in), ptr(self.emb_h), ptr(self.emb), self.B, eng.mc,
                                        self.emb.shape[1], eng.act, hip.stream()), "rho_timestep_embed")
        s = hip.stream()
        if self.drop_active:
            # a fresh stretch of every block's Philox stream for this forward (and its backward, which reads the same counter)
            check(self.L.rho_step_advance(None, ptr(self.drop_ctr), self.drop_delta, s), "rho_step_advance")
        for op in self.ops:
            rc = op(s)
            if rc != 0:
                check(rc, "UNet plan launch")
        return self.out

    def run_backward(self, dpred: Tensor, on_ready=None) -> None:
        eng = self.eng
        for p_ in eng.model.parameters():
            if p_.grad is None:
                p_.grad = torch.zeros_like(p_)
        self.dpred_in.copy_(dpred.reshape(self.dpred_in.shape))
        s = hip.stream()
        marks: Dict[int, list] = {}
        for i, ps in self.bwd_marks:
            marks.setdefault(i, []).extend(ps)
        for i, op in enumerate(self.bwd):
            rc = op(s)
            if rc != 0:
                check(rc, "UNet backward plan launch")
            if on_ready is not None and (i + 1) in marks:
                on_ready(marks[i + 1])
        if self.cond_dev is not None:
            cd = self.cond_dev
            gsig = tuple(w.grad.data_ptr() for w in cd["weights"])
            if getattr(self, "_gp_sig", None) != gsig:              # gradient storage is stable under the arena optimizer
                self._gp = torch.tensor(list(gsig), dtype=torch.int64, device=self.demb.device)
                self._gp_sig = gsig
            gp = self._gp
            check(self.L.rho_multi_embed_bwd(ptr(self.demb), ptr(self.cond_idx), ptr(gp), cd["nkeys"], self.B, self.demb.shape[1], s),
                  "rho_multi_embed_bwd")
        elif self.cond_src is not None:
            with torch.enable_grad():
                self.cond_src.backward(self.demb)      # a user-supplied cond_fn module
        if on_ready is not None:
            on_ready([])

    def profile(self, repeats: int = 3, backward: bool = False) -> List[dict]:
        """Replay the plan with a HIP event pair around every launch (events recorded on the stream the
        kernels are launched on) and return per-launch dicts {kind, flops, bytes, ms} (ms = mean over
        repeats).  Inputs are whatever the buffers currently hold."""
        s = hip.stream()
        lst, infos = (self.bwd, self.bwd_info) if backward else (self.ops, self.info)
        tot = [0.0] * len(lst)
        ov_was, self._overlap_on = getattr(self, "_overlap_on", False), False       # every launch on the timed stream, in order
        for _ in range(repeats):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in lst]
            for op, (e0, e1) in zip(lst, evs):
                e0.record()
                rc = op(s)
                e1.record()
                if rc != 0:
                    check(rc, "UNet plan launch (profile)")
            torch.cuda.synchronize()
            for i, (e0, e1) in enumerate(evs):
                tot[i] += e0.elapsed_time(e1)
        self._overlap_on = ov_was
        return [dict(info, ms=tot[i] / repeats) for i, info in enumerate(infos)]
