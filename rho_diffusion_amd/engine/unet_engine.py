"""Lowering of a ``UNet`` module tree (models/unet_v2.py) to a flat list of HIP kernel launches.

Reference semantics: UNet.forward, rho_diffusion/models/unet_v2.py:685-732, with ResBlock._forward
(:273-293), AttentionBlock._forward (:336-342), Downsample / Upsample (:103-169) and the final
GroupNorm-SiLU-conv (:679-683).

Per forward and per ResBlock the launches are:
    gn_partial, gn_finalize            statistics + folded (GroupNorm * FiLM) affine per (n, c)
    conv3   (prologue: affine+SiLU)    in_layers   [+ additive embedding in the epilogue]
    gn_partial, gn_finalize
    [conv1x1 skip]
    conv3   (prologue: affine+SiLU, epilogue: + skip)
torch.cat of the skip connections, nearest-upsample, strides, SiLU, FiLM and residual adds never
exist as separate passes over HBM.  A plan (all descriptors + all buffers) is built once per
(batch, spatial shape) and replayed; buffer addresses are stable so the replay can be captured in
a HIP graph.  PyTorch supplies memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import nn

from .. import hip
from ..hip import check, ptr
from . import ops

Tensor = torch.Tensor


class _ConvW:
    """A conv weight prepared for the kernel: [taps, coutp, cinp] in the engine dtype + padded fp32 bias."""

    def __init__(self, weight: nn.Parameter, bias: nn.Parameter, dtype, row_src: Optional[Tensor] = None,
                 cout_pad_to: int = 32):
        self.weight, self.bias_param, self.dtype = weight, bias, dtype
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        k = list(weight.shape[2:])
        while len(k) < 3:
            k.insert(0, 1)
        self.kernel = tuple(int(v) for v in k)
        ck = ops.elem_chunk(dtype)
        self.cinp = ((self.cin + ck - 1) // ck) * ck
        self.coutp = ((self.cout + cout_pad_to - 1) // cout_pad_to) * cout_pad_to
        self.row_src = row_src
        dev = weight.device
        self.w = torch.empty(int(k[0] * k[1] * k[2]), self.coutp, self.cinp, dtype=dtype, device=dev)
        self.b = torch.zeros(self.coutp, dtype=torch.float32, device=dev)
        self.refresh()

    def refresh(self) -> None:
        w = self.weight.detach()
        if not w.is_contiguous():
            w = w.contiguous()
        ops.prep_conv_weight(w, self.dtype, self.coutp, self.cinp, self.row_src, out=self.w)
        b = self.bias_param.detach()
        if self.row_src is not None:
            b = b[self.row_src.long()]           # gather (data movement only)
        self.b[: b.numel()].copy_(b)


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
        self._plans: Dict[tuple, "_Plan"] = {}
        self._convs: List[_ConvW] = []
        self._conv_of: Dict[int, _ConvW] = {}
        self._film_blocks: List[nn.Module] = []
        self._sin_table: Optional[Tensor] = None
        self._sin_rows = 0
        self._param_version = -1
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
                self._conv(mod.op)
            elif isinstance(mod, Upsample):
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

    def refresh_weights(self, force: bool = False) -> None:
        """Re-run the weight preparation kernels if any parameter changed (optimizer step, load_state_dict)."""
        v = self._versions()
        if not force and v == self._param_version:
            return
        for cw in self._convs:
            cw.refresh()
        off = 0
        for blk in self._film_blocks:
            lin = blk.emb_layers[1]
            n = lin.weight.shape[0]
            self.film_w[off:off + n].copy_(lin.weight.detach())
            self.film_b[off:off + n].copy_(lin.bias.detach())
            off += n
        self._param_version = v

    def sin_table(self, rows: int) -> Tensor:
        """Rows t = 0..rows-1 of the interleaved sin/cos embedding (models/common.py), built on the
        host exactly as the reference evaluates it and gathered on the device per step."""
        if self._sin_table is None or self._sin_rows < rows:
            from ..models.common import sinosoidal_position_embedding
            rows = max(rows, 1024)
            tab = sinosoidal_position_embedding(torch.arange(rows), self.mc)
            self._sin_table = tab.to(self.device).contiguous()
            self._sin_rows = rows
        return self._sin_table

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, timesteps: Tensor, y: Optional[Tensor] = None,
                t_scalar_dev: Optional[Tensor] = None) -> Tensor:
        hip.require_gpu(x, "x")
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        key = (tuple(x.shape), y is not None)
        plan = self._plans.get(key)
        if plan is None:
            with torch.inference_mode(False):     # plan buffers must stay ordinary tensors
                plan = self._plans[key] = _Plan(self, tuple(x.shape), y is not None)
        self.refresh_weights()
        return plan.run(x, timesteps, y, t_scalar_dev)


class _Plan:
    """All buffers + launch closures for one input shape."""

    def __init__(self, eng: UNetEngine, xshape: Tuple[int, ...], has_y: bool):
        from ..models.unet_v2 import AttentionBlock, Downsample, ResBlock, Upsample
        self.eng = eng
        m = eng.model
        dt = eng.dtype
        dev = eng.device
        self.ops: List[Callable[[int], int]] = []
        self.info: List[dict] = []     # per launch: kind, algorithmic flops / bytes (for bench roofline)
        self.keep: List[object] = []   # descriptors / tensors referenced by raw pointers
        L = hip.lib()
        B = xshape[0]
        self.B = B
        self.xshape = xshape
        D, H, W = ops.spatial5(xshape[2:])
        dims = eng.dims
        e = 4 * eng.mc

        def buf(*shape, dtype=dt):
            t = torch.empty(*shape, dtype=dtype, device=dev)
            self.keep.append(t)
            return t

        # ---- embedding chain: table gather -> Linear -> SiLU -> Linear (+cond) -> batched FiLM GEMV
        self.sin_in = buf(B, eng.mc, dtype=torch.float32)
        self.emb_h = buf(B, e, dtype=torch.float32)
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

        op_linear(self.sin_in, te0.weight, te0.bias, None, self.emb_h, False, True)
        op_linear(self.emb_h, te2.weight, te2.bias, self.cond, self.emb, False, False)
        if eng.film_total:
            op_linear(self.emb, eng.film_w, eng.film_b, None, self.film, True, False)

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
            if film_blk is not None:
                off = eng._film_off[id(film_blk)]
                scale = self.film.data_ptr() + 4 * off
                shift = self.film.data_ptr() + 4 * (off + Cc)
                stride = self.film.shape[1]
            a1 = (ptr(x1), c1, ptr(x2), c2, hip.dtype_code(dt), N, S, ptr(part))
            a2 = (ptr(part), N, Cc, S, nblk, ptr(norm.weight), ptr(norm.bias), scale, shift, stride, ptr(st), ptr(a), ptr(b))
            esz = 2 if dt == torch.bfloat16 else 4
            self.ops.append(lambda s, a=a1: L.rho_gn_partial(*a, s))
            self.info.append(dict(kind="gn_partial", flops=3.0 * N * S * Cc, bytes=float(esz) * N * S * Cc))
            self.ops.append(lambda s, a=a2: L.rho_gn_finalize(*a, s))
            self.info.append(dict(kind="gn_finalize", flops=0.0, bytes=4.0 * N * Cc * 4))
            return a, b

        def conv(x1, x2, cw, *, stride_hw=(1, 1), up_hw=(0, 0), pre=None, pre_silu=False, res=None, res_add=None,
                 res_add_stride=0, split=None, y2_dtype=None):
            cout = cw.cout
            split_ = cout if split is None else split
            N, Do, Ho, Wo = ops.conv_out_shape(x1.shape, cw.kernel, stride_hw, up_hw)
            y = buf(N, Do, Ho, Wo, split_) if split_ > 0 else None
            y2 = buf(N, cout - split_, Do * Ho * Wo, dtype=y2_dtype or dt) if split_ < cout else None
            d = ops.make_conv_desc(x1, x2, cw.w, cw.b, kernel=cw.kernel, cout=cout, split=split_, y=y, y2=y2,
                                   stride_hw=stride_hw, up_hw=up_hw, pre_a=pre[0] if pre else None,
                                   pre_b=pre[1] if pre else None, pre_silu=pre_silu, res=res, res_add=None,
                                   res_add_stride=res_add_stride)
            if res_add is not None:
                d.res_add = res_add
            self.keep.append(d)
            self.ops.append(lambda s, d=d: L.rho_conv_nd_fwd(C.byref(d), s))
            taps = cw.kernel[0] * cw.kernel[1] * cw.kernel[2]
            esz = 2 if dt == torch.bfloat16 else 4
            npos_out = N * Do * Ho * Wo
            npos_in = x1.numel() // x1.shape[-1]
            self.info.append(dict(
                kind="conv3" if taps > 1 else "conv1", taps=taps, cin=cw.cin, cout=cout, positions=npos_out,
                flops=2.0 * npos_out * cout * cw.cin * taps,                       # algorithmic (unpadded) MACs * 2
                bytes=float(esz) * (npos_in * cw.cin + npos_out * cout * (2 if res is not None else 1) + taps * cout * cw.cin)))
            return y, y2

        def resblock(blk, h1, h2):
            a1, b1 = gn(h1, h2, blk.in_layers[0])
            radd, rstride = None, 0
            if not blk.use_scale_shift_norm:
                radd = self.film.data_ptr() + 4 * eng._film_off[id(blk)]
                rstride = self.film.shape[1]
            t1, _ = conv(h1, h2, eng._conv(blk.in_layers[2]), pre=(a1, b1), pre_silu=True, res_add=radd, res_add_stride=rstride)
            a2, b2 = gn(t1, None, blk.out_layers[0], film_blk=blk if blk.use_scale_shift_norm else None)
            if isinstance(blk.skip_connection, nn.Identity):
                assert h2 is None
                sk = h1
            else:
                sk, _ = conv(h1, h2, eng._conv(blk.skip_connection))
            out, _ = conv(t1, None, eng._conv(blk.out_layers[3]), pre=(a2, b2), pre_silu=True, res=sk)
            return out

        def attention(blk, xin):
            N, Dd, Hh, Ww, Cc = xin.shape
            T = Dd * Hh * Ww
            a, b = gn(xin, None, blk.norm)
            qk, vt = conv(xin, None, eng._conv(blk.qkv), pre=(a, b), pre_silu=False, split=2 * Cc)
            ao = buf(N, Dd, Hh, Ww, Cc)
            args = (ptr(qk), ptr(vt), ptr(ao), None, hip.dtype_code(dt), N, T, blk.num_heads, Cc // blk.num_heads)
            self.ops.append(lambda s, a=args: L.rho_attention_fwd(*a, s))
            esz = 2 if dt == torch.bfloat16 else 4
            self.info.append(dict(kind="attention", flops=4.0 * N * T * T * Cc, bytes=float(esz) * 4 * N * T * Cc))
            out, _ = conv(ao, None, eng._conv(blk.proj_out), res=xin)
            return out

        def run_block(seq, h1, h2):
            for layer in seq:
                if isinstance(layer, ResBlock):
                    h1, h2 = resblock(layer, h1, h2), None
                elif isinstance(layer, AttentionBlock):
                    h1 = attention(layer, h1)
                elif isinstance(layer, Downsample):
                    st = (2, 2) if dims >= 2 else (1, 2)
                    h1, _ = conv(h1, None, eng._conv(layer.op), stride_hw=st)
                elif isinstance(layer, Upsample):
                    up = (1, 1) if dims >= 2 else (0, 1)
                    h1, _ = conv(h1, None, eng._conv(layer.conv), up_hw=up)
                else:  # the stem conv
                    h1, _ = conv(h1, None, eng._conv(layer))
            return h1

        # ---- the network
        stem = eng._conv(m.input_blocks[0][0])
        self.x_in = buf(*xshape, dtype=torch.float32)
        self.x_cl = buf(B, D, H, W, stem.cinp)
        pk = (ptr(self.x_in), ptr(self.x_cl), hip.dtype_code(dt), B, xshape[1], D * H * W, stem.cinp)
        self.ops.append(lambda s, a=pk: L.rho_pack_input(*a, s))
        self.info.append(dict(kind="pack", flops=0.0, bytes=4.0 * B * xshape[1] * D * H * W + 2.0 * B * D * H * W * stem.cinp))

        hs = []
        h = self.x_cl
        for blk in m.input_blocks:
            h = run_block(blk, h, None)
            hs.append(h)
        h = run_block(m.middle_block, h, None)
        for blk in m.output_blocks:
            h = run_block(blk, h, hs.pop())
        a, b = gn(h, None, m.out[0])
        _, y2 = conv(h, None, eng._conv(m.out[2]), pre=(a, b), pre_silu=True, split=0, y2_dtype=torch.float32)
        self.out = y2.view(B, m.out_channels, *xshape[2:])

    def run(self, x: Tensor, timesteps: Optional[Tensor], y: Optional[Tensor], t_scalar_dev: Optional[Tensor]) -> Tensor:
        eng = self.eng
        m = eng.model
        if x.data_ptr() != self.x_in.data_ptr():
            self.x_in.copy_(x)
        # timestep embedding rows (interleaved sin/cos), gathered on the device
        if t_scalar_dev is not None:
            ops.embed_gather(eng.sin_table(1024), None, self.B, t_scalar_dev, out=self.sin_in)
        else:
            hip.require_gpu(timesteps, "timesteps")
            ts = timesteps.to(torch.int64).contiguous()
            ops.embed_gather(eng.sin_table(1024), ts, self.B, out=self.sin_in)
        if self.cond is not None:
            # label handling of unet_v2.py:702-719
            if y.dim() == 2 and tuple(y.shape) == tuple(self.emb.shape):
                self.cond.copy_(y.to(self.cond.device))
            else:
                if y.dim() == 1:
                    assert y.shape == (x.shape[0],)
                else:
                    assert y.shape[0] == self.emb.shape[0]
                self.cond.copy_(m.cond_fn(y))
        s = hip.stream()
        for op in self.ops:
            rc = op(s)
            if rc != 0:
                check(rc, "UNet plan launch")
        return self.out

    def profile(self, repeats: int = 3) -> List[dict]:
        """Replay the plan with a HIP event pair around every launch (events recorded on the stream
        the kernels are launched on) and return per-launch dicts {kind, flops, bytes, ms} (ms = mean over repeats).
        Inputs are whatever the buffers currently hold."""
        s = hip.stream()
        tot = [0.0] * len(self.ops)
        for _ in range(repeats):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in self.ops]
            for op, (e0, e1) in zip(self.ops, evs):
                e0.record()
                rc = op(s)
                e1.record()
                if rc != 0:
                    check(rc, "UNet plan launch (profile)")
            torch.cuda.synchronize()
            for i, (e0, e1) in enumerate(evs):
                tot[i] += e0.elapsed_time(e1)
        return [dict(info, ms=tot[i] / repeats) for i, info in enumerate(self.info)]
