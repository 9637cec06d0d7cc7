"""Tensor-level wrappers over the C ABI (include/rho_hip.h): shape/dtype checks on the host,
raw device pointers into the kernels.  All functions enqueue on the current torch stream and
return torch tensors that own the device memory.  No arithmetic happens in PyTorch here."""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence, Tuple

import torch

from .. import hip
from ..hip import ConvDesc, RhoHipError, check, dtype_code, ptr, stream

Tensor = torch.Tensor


def _f32c(t: Tensor, name: str) -> Tensor:
    hip.require_gpu(t, name)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RhoHipError(f"{name} must be contiguous float32, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


def spatial5(shape: Sequence[int]) -> Tuple[int, int, int]:
    """(D, H, W) of a 1/2/3-D spatial shape (missing leading axes are 1)."""
    s = list(shape)
    while len(s) < 3:
        s.insert(0, 1)
    return int(s[0]), int(s[1]), int(s[2])


def elem_chunk(dtype: torch.dtype) -> int:
    """channels per 64-byte chunk: the conv kernel's channel granularity."""
    return 32 if dtype == torch.bfloat16 else 16


# ----------------------------------------------------------------------------- diffusion loops
def q_sample(x0: Tensor, eps: Tensor, t: Tensor, alpha_bar: Tensor, out: Optional[Tensor] = None,
             nan_flag: Optional[Tensor] = None) -> Tensor:
    x0, eps, alpha_bar = _f32c(x0, "x0"), _f32c(eps, "eps"), _f32c(alpha_bar, "alpha_bar")
    if t.dtype != torch.int64 or not t.is_cuda:
        raise RhoHipError("t must be an int64 GPU tensor")
    out = torch.empty_like(x0) if out is None else out
    B = x0.shape[0]
    check(hip.lib().rho_q_sample(ptr(x0), ptr(eps), ptr(out), ptr(alpha_bar), ptr(t), B, x0.numel() // B, alpha_bar.numel(),
                                 ptr(nan_flag), stream()), "rho_q_sample")
    return out


def p_sample_step(x: Tensor, eps_hat: Tensor, z: Optional[Tensor], coef_table: Tensor, t_dev: Tensor) -> Tensor:
    _f32c(x, "x"), _f32c(eps_hat, "eps_hat"), _f32c(coef_table, "coef_table")
    if t_dev.dtype != torch.int32:
        raise RhoHipError("t_dev must be int32[1] on the GPU")
    check(hip.lib().rho_p_sample_step(ptr(x), ptr(eps_hat), ptr(z), ptr(coef_table), ptr(t_dev), x.numel(), stream()),
          "rho_p_sample_step")
    return x


def q_sample_coef(x0: Tensor, eps: Tensor, t: Tensor, coef_a: Tensor, coef_b: Tensor, out: Optional[Tensor] = None,
                  err_flag: Optional[Tensor] = None) -> Tensor:
    """x_t = a[t] * x0 + b[t] * eps with explicit float32 tables (GaussianDiffusionPipeline.q_sample)."""
    x0, eps, coef_a, coef_b = _f32c(x0, "x0"), _f32c(eps, "eps"), _f32c(coef_a, "coef_a"), _f32c(coef_b, "coef_b")
    if t.dtype != torch.int64 or not t.is_cuda:
        raise RhoHipError("t must be an int64 GPU tensor")
    out = torch.empty_like(x0) if out is None else out
    B = x0.shape[0]
    check(hip.lib().rho_q_sample_coef(ptr(x0), ptr(eps), ptr(out), ptr(coef_a), ptr(coef_b), ptr(t), B, x0.numel() // B,
                                      min(coef_a.numel(), coef_b.numel()), ptr(err_flag), stream()), "rho_q_sample_coef")
    return out


def abs_quantile(x: Tensor, q: float, out: Optional[Tensor] = None, workspace: Optional[Tensor] = None) -> Tensor:
    """Per-sample torch.quantile(|x|.flatten(1), q) (float32, linear interpolation), exact radix select on the device."""
    x = _f32c(x, "x")
    B = x.shape[0]
    need = hip.lib().rho_abs_quantile_workspace_bytes(B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.int32, device=x.device)
    out = torch.empty(B, dtype=torch.float32, device=x.device) if out is None else out
    check(hip.lib().rho_abs_quantile(ptr(x), B, x.numel() // B, float(q), ptr(workspace), ptr(out), stream()), "rho_abs_quantile")
    return out


def ddim_step(x_t: Tensor, model_out: Tensor, quantile: Tensor, noise: Optional[Tensor], x_prev: Tensor, pred_xstart: Optional[Tensor],
              c_recip: float, c_recipm1: float, sqrt_abar_prev: float, coef_eps: float, sigma_masked: float) -> Tensor:
    _f32c(x_t, "x_t"), _f32c(model_out, "model_out"), _f32c(quantile, "quantile"), _f32c(x_prev, "x_prev")
    B = x_t.shape[0]
    check(hip.lib().rho_ddim_step(ptr(x_t), ptr(model_out), ptr(quantile), ptr(noise), ptr(x_prev), ptr(pred_xstart), B,
                                  x_t.numel() // B, c_recip, c_recipm1, sqrt_abar_prev, coef_eps, sigma_masked, stream()),
          "rho_ddim_step")
    return x_prev


def step_advance(t_dev: Optional[Tensor], offset_dev: Optional[Tensor], delta: int) -> None:
    check(hip.lib().rho_step_advance(ptr(t_dev), ptr(offset_dev), delta, stream()), "rho_step_advance")


def philox_normal(out: Tensor, seed: int, offset: int = 0, offset_dev: Optional[Tensor] = None) -> Tensor:
    _f32c(out, "out")
    check(hip.lib().rho_philox_normal(ptr(out), out.numel(), seed & (2 ** 64 - 1), offset, ptr(offset_dev), stream()),
          "rho_philox_normal")
    return out


def mse(a: Tensor, b: Tensor, want_grad: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
    _f32c(a, "a"), _f32c(b, "b")
    buf = torch.empty(1 + 1024, dtype=torch.float32, device=a.device)        # loss + the ordered reduction's block partials
    loss = buf[:1]
    grad = torch.empty_like(a) if want_grad else None
    check(hip.lib().rho_mse_ws(ptr(a), ptr(b), ptr(loss), ptr(grad), a.numel(), buf.data_ptr() + 4, 1024, stream()), "rho_mse_ws")
    return loss, grad


def mean_flat(x: Tensor) -> Tensor:
    """Mean over all non-batch axes (layers.py:105-110) on the device, float32, fixed summation order."""
    x = _f32c(x, "x")
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(hip.lib().rho_mean_flat(ptr(x), ptr(out), x.shape[0], x.numel() // x.shape[0], stream()), "rho_mean_flat")
    return out


def deterministic() -> bool:
    """The library's reproducibility switch (RHO_DETERMINISTIC=1 / rho_set_deterministic)."""
    return bool(hip.lib().rho_get_deterministic())


def set_deterministic(on: bool) -> bool:
    return bool(hip.lib().rho_set_deterministic(1 if on else 0))


def adamw(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, beta1: float, beta2: float, eps: float,
          weight_decay: float, step: int) -> None:
    for name, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _f32c(t, name)
    check(hip.lib().rho_adamw(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step, stream()),
          "rho_adamw")


# ----------------------------------------------------------------------------- embeddings
# activation codes of the C ABI (include/rho_hip.h, rho_timestep_embed): the elementwise, parameter-free entries of the reference's
# activation registry (rho_diffusion/registry.py:162-170)
ACT_CODES = {"Identity": 0, "SiLU": 1, "ReLU": 2, "GELU": 3, "Tanh": 4, "Sigmoid": 5, "ELU": 6}


def sinusoid_frequencies(dim: int, wavelength: int = 10000, device=None) -> Tensor:
    """omega_i = wavelength^(2i / dim), i = 0 .. dim/2 - 1, float32: the denominators of the timestep sinusoid
    (models/common.py:38 evaluates this very expression; the kernel divides t by it)."""
    if dim % 2:
        raise RhoHipError("`dim` should be dividable by 2.")
    i = torch.arange(dim // 2)
    return torch.pow(wavelength, 2 * i / dim).to(device=device, dtype=torch.float32).contiguous()


def timestep_embed(omega: Tensor, t: Optional[Tensor], batch: int, *, t_scalar_dev: Optional[Tensor] = None,
                   w0: Optional[Tensor] = None, b0: Optional[Tensor] = None, w2: Optional[Tensor] = None,
                   b2: Optional[Tensor] = None, cond: Optional[Tensor] = None, pe_out: Optional[Tensor] = None,
                   h_out: Optional[Tensor] = None, emb_out: Optional[Tensor] = None, act: int = 1):
    """Sinusoid of any integer t (+ the time_embed MLP and the label-embedding add when w0 is given): rho_timestep_embed.
    ``act``: activation code between the two linears (1 = SiLU, see ACT_CODES)."""
    _f32c(omega, "omega")
    dim = 2 * omega.numel()
    if t is not None and (t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous()):
        raise RhoHipError("t must be a contiguous int64 GPU tensor")
    edim = w0.shape[0] if w0 is not None else 0
    if pe_out is None:
        pe_out = torch.empty(batch, dim, dtype=torch.float32, device=omega.device)
    if w0 is not None and emb_out is None:
        emb_out = torch.empty(batch, edim, dtype=torch.float32, device=omega.device)
    check(hip.lib().rho_timestep_embed(ptr(omega), ptr(t), ptr(t_scalar_dev), ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(cond),
                                       ptr(pe_out), ptr(h_out), ptr(emb_out), batch, dim, edim, int(act), stream()), "rho_timestep_embed")
    return pe_out if w0 is None else emb_out


def randint(n: int, high: int, seed: int, offset: int = 0, offset_dev: Optional[Tensor] = None, out: Optional[Tensor] = None,
            device=None) -> Tensor:
    """Uniform int64 on [0, high) from the Philox stream (seed, offset): random_timesteps on the device."""
    out = torch.empty(n, dtype=torch.int64, device=device) if out is None else out
    hip.require_gpu(out, "out")
    check(hip.lib().rho_randint(ptr(out), n, high, seed & (2 ** 64 - 1), offset, ptr(offset_dev), stream()), "rho_randint")
    return out


def sph_harm_fields(lm: Tensor, grid: int, out: Optional[Tensor] = None, minmax_in: Optional[Tensor] = None,
                    minmax_out: Optional[Tensor] = None) -> Tensor:
    """Spherical-harmonic density fields float32 [B, G, G, G] for (l, m) = lm[b] (int32 [B, 2] on the GPU)."""
    hip.require_gpu(lm, "lm")
    if lm.dtype != torch.int32 or lm.dim() != 2 or lm.shape[1] != 2 or not lm.is_contiguous():
        raise RhoHipError("lm must be a contiguous int32 [B, 2] tensor")
    B = lm.shape[0]
    out = torch.empty(B, grid, grid, grid, dtype=torch.float32, device=lm.device) if out is None else out
    ws = torch.empty(int(hip.lib().rho_sph_harm_workspace_bytes(B, grid)) // 8, dtype=torch.float64, device=lm.device)
    for name, mm in (("minmax_in", minmax_in), ("minmax_out", minmax_out)):
        if mm is not None and (mm.dtype != torch.float64 or tuple(mm.shape) != (B, 4) or not mm.is_cuda or not mm.is_contiguous()):
            raise RhoHipError(f"{name} must be a contiguous float64 [B, 4] GPU tensor")
    check(hip.lib().rho_sph_harm_fields(ptr(lm), B, grid, ptr(out), ptr(ws), ptr(minmax_in), ptr(minmax_out), stream()),
          "rho_sph_harm_fields")
    return out


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor], add: Optional[Tensor] = None, act_in: bool = False,
           act_out: bool = False, out: Optional[Tensor] = None) -> Tensor:
    _f32c(x, "x"), _f32c(w, "w")
    B, K = x.shape
    O = w.shape[0]
    if w.shape[1] != K:
        raise RhoHipError(f"linear: weight {tuple(w.shape)} does not match input {tuple(x.shape)}")
    out = torch.empty(B, O, dtype=torch.float32, device=x.device) if out is None else out
    check(hip.lib().rho_linear(ptr(x), ptr(w), ptr(bias), ptr(add), ptr(out), B, K, O, int(act_in), int(act_out), stream()),
          "rho_linear")
    return out


# ----------------------------------------------------------------------------- layout
def pack_input(x: Tensor, dtype: torch.dtype, cpad: Optional[int] = None, out: Optional[Tensor] = None) -> Tensor:
    """[N, C, *S] float32 -> channels-last [N, D, H, W, Cpad] in the engine dtype."""
    _f32c(x, "x")
    N, Cc = x.shape[0], x.shape[1]
    D, H, W = spatial5(x.shape[2:])
    ck = elem_chunk(dtype)
    cpad = cpad or ((Cc + ck - 1) // ck) * ck
    out = torch.empty(N, D, H, W, cpad, dtype=dtype, device=x.device) if out is None else out
    check(hip.lib().rho_pack_input(ptr(x), ptr(out), dtype_code(dtype), N, Cc, D * H * W, cpad, stream()), "rho_pack_input")
    return out


def prep_conv_weight(w: Tensor, dtype: torch.dtype, coutp: Optional[int] = None, cinp: Optional[int] = None,
                     row_src: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """PyTorch conv weight [Cout, Cin, *k] float32 -> [taps, CoutP, CinP] in the engine dtype."""
    _f32c(w, "w")
    cout, cin = w.shape[0], w.shape[1]
    taps = int(math.prod(w.shape[2:])) if w.dim() > 2 else 1
    ck = elem_chunk(dtype)
    cinp = cinp or ((cin + ck - 1) // ck) * ck
    coutp = coutp or ((cout + 31) // 32) * 32
    out = torch.empty(taps, coutp, cinp, dtype=dtype, device=w.device) if out is None else out
    if row_src is not None and (row_src.dtype != torch.int32 or row_src.numel() != coutp):
        raise RhoHipError("row_src must be int32[coutp]")
    check(hip.lib().rho_prep_conv_weight(ptr(w), ptr(out), dtype_code(dtype), cout, cin, taps, coutp, cinp, ptr(row_src), stream()),
          "rho_prep_conv_weight")
    return out


def prep_conv_weight_phase(w: Tensor, dtype: torch.dtype, phase_hw: Tuple[int, int], out: Optional[Tensor] = None,
                           dgrad: bool = False) -> Tensor:
    """Weights of one sub-pixel phase of a conv behind a nearest x2 upsample (rho_conv_desc.ph_h / ph_w): parameter
    [Cout, Cin, *k] (k = 3 on a phased axis) -> [kd * kh' * kw', CoutP, CinP] with kh' / kw' = 2 on the phased axes; dgrad: the
    data-gradient layout [taps, ceil32(Cin), ceilCK(Cout)] (flipped, transposed) for a launch with phd_h / phd_w."""
    _f32c(w, "w")
    cout, cin = w.shape[0], w.shape[1]
    k = [1] * (5 - w.dim()) + [int(v) for v in w.shape[2:]]
    k2 = (k[0], 2 if phase_hw[0] else k[1], 2 if phase_hw[1] else k[2])
    ck = elem_chunk(dtype)
    if dgrad:
        cinp = ((cin + 31) // 32) * 32                   # rows of the dgrad weights = its output channels
        coutp = ((cout + ck - 1) // ck) * ck             # columns = its input channels (chunked)
        shape = (k2[0] * k2[1] * k2[2], cinp, coutp)
    else:
        cinp = ((cin + ck - 1) // ck) * ck
        coutp = ((cout + 31) // 32) * 32
        shape = (k2[0] * k2[1] * k2[2], coutp, cinp)
    out = torch.empty(*shape, dtype=dtype, device=w.device) if out is None else out
    check(hip.lib().rho_prep_conv_weight_phase(ptr(w), ptr(out), dtype_code(dtype), cout, cin, k[0], k[1], k[2], int(phase_hw[0]),
                                               int(phase_hw[1]), coutp, cinp, int(dgrad), stream()), "rho_prep_conv_weight_phase")
    return out


def prep_conv_weight_sel(w: Tensor, dtype: torch.dtype, sel_hw, *, flip_d: bool = False, dgrad: bool = False,
                         out: Optional[Tensor] = None) -> Tensor:
    """Tap selection of a 3-tap parameter for the parity splits of a stride-2 conv (rho_prep_conv_weight_sel): sel_hw = (taps of H,
    taps of W), each a tuple of source tap indices (None keeps the axis)."""
    _f32c(w, "w")
    cout, cin = w.shape[0], w.shape[1]
    k = [1] * (5 - w.dim()) + [int(v) for v in w.shape[2:]]
    sh = tuple(range(k[1])) if sel_hw[0] is None else tuple(sel_hw[0])
    sw = tuple(range(k[2])) if sel_hw[1] is None else tuple(sel_hw[1])
    code = lambda t: sum(int(v) << (4 * i) for i, v in enumerate(t))      # noqa: E731
    ck = elem_chunk(dtype)
    taps2 = k[0] * len(sh) * len(sw)
    if dgrad:
        cinp, coutp = ((cin + 31) // 32) * 32, ((cout + ck - 1) // ck) * ck
        shape = (taps2, cinp, coutp)
    else:
        cinp, coutp = ((cin + ck - 1) // ck) * ck, ((cout + 31) // 32) * 32
        shape = (taps2, coutp, cinp)
    out = torch.empty(*shape, dtype=dtype, device=w.device) if out is None else out
    check(hip.lib().rho_prep_conv_weight_sel(ptr(w), ptr(out), dtype_code(dtype), cout, cin, k[0], k[1], k[2], len(sh), len(sw), code(sh),
                                             code(sw), int(flip_d), coutp, cinp, int(dgrad), stream()), "rho_prep_conv_weight_sel")
    return out


class PrepTable:
    """Device table of rho_prep_op for rho_prep_batch: every prepared weight layout / padded bias / fp32 copy of a model in one
    launch.  ``add_*`` mirror the per-tensor prep calls (same arguments, same results); ``launch()`` uploads the table once."""

    def __init__(self, device):
        self.device = device
        self.ops: list = []
        self.keep: list = []
        self._dev = None
        self._blocks = 0

    def _push(self, kind, w: Tensor, out: Tensor, *, cout, cin, d1, d2, perm=None, k=(1, 1, 1), kh2=0, kw2=0, ph=(0, 0), sel=(0, 0),
              flip_d=False, dgrad=False):
        _f32c(w, "w")
        op = hip.PrepOp()
        op.w, op.out, op.perm = w.data_ptr(), out.data_ptr(), (perm.data_ptr() if perm is not None else None)
        op.cout, op.cin, op.d1, op.d2, op.total = int(cout), int(cin), int(d1), int(d2), int(out.numel())
        op.kind, op.dtype = kind, hip.dtype_code(out.dtype)
        op.kd, op.kh, op.kw, op.kh2, op.kw2 = int(k[0]), int(k[1]), int(k[2]), int(kh2), int(kw2)
        op.ph_h, op.ph_w, op.sel_h, op.sel_w, op.flip_d, op.dgrad = int(ph[0]), int(ph[1]), int(sel[0]), int(sel[1]), int(flip_d), int(dgrad)
        nblk = max(1, min((op.total + 1023) // 1024, int(os.environ.get("RHO_PREP_MAX_BLOCKS", "2048"))))
        op.blk0, op.nblk = self._blocks, nblk
        self._blocks += nblk
        self.ops.append(op)
        self.keep.append((w, out, perm))
        self._dev = None

    @staticmethod
    def _k3(w: Tensor):
        return [1] * (5 - w.dim()) + [int(v) for v in w.shape[2:]] if w.dim() > 2 else [1, 1, 1]

    def add_fwd(self, w: Tensor, out: Tensor, row_src: Optional[Tensor] = None, k=None):
        """rho_prep_conv_weight: [Cout, Cin, *k] -> out [taps, CoutP, CinP].  ``k`` overrides the kernel extents of ``w``'s shape
        (a parameter re-read as a GEMM: any (kd, kh, kw) whose product is the tap count of the view)."""
        k = self._k3(w) if k is None else k
        self._push(hip.PREP_FWD, w, out, cout=w.shape[0], cin=w.shape[1], d1=out.shape[1], d2=out.shape[2], perm=row_src, k=k)

    def add_dgrad(self, w: Tensor, out: Tensor, col_src: Optional[Tensor] = None):
        self._push(hip.PREP_DGRAD, w, out, cout=w.shape[0], cin=w.shape[1], d1=out.shape[1], d2=out.shape[2], perm=col_src, k=self._k3(w))

    def add_phase(self, w: Tensor, out: Tensor, phase_hw, dgrad: bool = False):
        self._push(hip.PREP_PHASE, w, out, cout=w.shape[0], cin=w.shape[1], d1=out.shape[1], d2=out.shape[2], k=self._k3(w), ph=phase_hw,
                   dgrad=dgrad)

    def add_sel(self, w: Tensor, out: Tensor, sel_hw, flip_d: bool = False, dgrad: bool = False):
        k = self._k3(w)
        sh = tuple(range(k[1])) if sel_hw[0] is None else tuple(sel_hw[0])
        sw = tuple(range(k[2])) if sel_hw[1] is None else tuple(sel_hw[1])
        code = lambda t: sum(int(v) << (4 * i) for i, v in enumerate(t))      # noqa: E731
        self._push(hip.PREP_SEL, w, out, cout=w.shape[0], cin=w.shape[1], d1=out.shape[1], d2=out.shape[2], k=k, kh2=len(sh), kw2=len(sw),
                   sel=(code(sh), code(sw)), flip_d=flip_d, dgrad=dgrad)

    def add_vec(self, src: Tensor, out: Tensor, perm: Optional[Tensor] = None, n: Optional[int] = None):
        """out.flat[i] = src.flat[perm[i] if perm else i] for i < n (default: all of src; perm[i] < 0: zero), zeros up to out.numel();
        src float32, out float32 or bf16 - a general gather (padded biases, plain copies, layouts no other kind describes)."""
        if not out.is_contiguous():
            raise RhoHipError("add_vec: out must be contiguous")
        n = src.numel() if n is None else n
        self._push(hip.PREP_VEC, src, out, cout=n, cin=src.numel(), d1=out.numel(), d2=1, perm=perm)

    def launch(self) -> None:
        if not self.ops:
            return
        if self._dev is None:
            raw = b"".join(bytes(op) for op in self.ops)
            self._dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        check(hip.lib().rho_prep_batch(self._dev.data_ptr(), len(self.ops), self._blocks, stream()), "rho_prep_batch")


# ----------------------------------------------------------------------------- GroupNorm
def gn_nblk(s: int) -> int:
    return int(hip.lib().rho_gn_nblk(s))


def gn_coeffs(x1: Tensor, x2: Optional[Tensor], gamma: Tensor, beta: Tensor, scale: Optional[Tensor] = None,
              shift: Optional[Tensor] = None, film_stride: int = 0, partials: Optional[Tensor] = None,
              a: Optional[Tensor] = None, b: Optional[Tensor] = None, stats: Optional[Tensor] = None):
    """GroupNorm(32) statistics of the (virtually concatenated) channels-last input and the folded
    per-(sample, channel) affine (a, b) consumed by the conv prologue."""
    N = x1.shape[0]
    c1 = x1.shape[-1]
    c2 = x2.shape[-1] if x2 is not None else 0
    S = x1.numel() // (N * c1)
    Cc = c1 + c2
    nblk = gn_nblk(S)
    dev = x1.device
    partials = torch.empty(N * nblk * (Cc // 8) * 16, dtype=torch.float32, device=dev) if partials is None else partials
    a = torch.empty(N, Cc, dtype=torch.float32, device=dev) if a is None else a
    b = torch.empty(N, Cc, dtype=torch.float32, device=dev) if b is None else b
    stats = torch.empty(N, 32, 2, dtype=torch.float32, device=dev) if stats is None else stats
    L = hip.lib()
    check(L.rho_gn_partial(ptr(x1), c1, ptr(x2), c2, dtype_code(x1.dtype), N, S, ptr(partials), stream()), "rho_gn_partial")
    check(L.rho_gn_finalize(ptr(partials), N, Cc, S, nblk, ptr(gamma), ptr(beta), ptr(scale), ptr(shift), film_stride,
                            ptr(stats), ptr(a), ptr(b), stream()), "rho_gn_finalize")
    return a, b, stats


# ----------------------------------------------------------------------------- convolution
def make_conv_desc(x1: Tensor, x2: Optional[Tensor], w: Tensor, bias: Tensor, *, kernel: Tuple[int, int, int],
                   cout: int, split: int, y: Optional[Tensor], y2: Optional[Tensor], stride_hw=(1, 1), up_hw=(0, 0),
                   pre_a: Optional[Tensor] = None, pre_b: Optional[Tensor] = None, pre_silu: bool = False,
                   res: Optional[Tensor] = None, res_add: Optional[Tensor] = None, res_add_stride: int = 0,
                   y2_cl: bool = False, res2: Optional[Tensor] = None, zs_hw=(0, 0), out_hw=(0, 0), phase_hw=(0, 0),
                   phase_dgrad_hw=(0, 0), skip: Optional[tuple] = None) -> ConvDesc:
    N, D, H, W, c1 = x1.shape
    d = ConvDesc()
    d.x1, d.x2 = ptr(x1), ptr(x2)
    d.pre_a, d.pre_b = ptr(pre_a), ptr(pre_b)
    d.w, d.bias = ptr(w), ptr(bias)
    d.res, d.res_add = ptr(res), ptr(res_add)
    d.y, d.y2 = ptr(y), ptr(y2)
    d.dtype = dtype_code(x1.dtype)
    d.y2_f32 = int(y2 is not None and y2.dtype == torch.float32)
    d.c1 = c1
    d.c2 = x2.shape[-1] if x2 is not None else 0
    d.cout, d.coutp, d.split = cout, w.shape[1], split
    d.n, d.d, d.h, d.w_ = N, D, H, W
    d.kd, d.kh, d.kw = kernel
    d.sh, d.sw = stride_hw
    d.up_h, d.up_w = up_hw
    d.pre_silu = int(pre_silu)
    d.res_add_stride = res_add_stride
    d.y2_cl = int(y2_cl)
    d.res2 = ptr(res2)
    d.zs_h, d.zs_w = int(zs_hw[0]), int(zs_hw[1])
    d.out_h, d.out_w = int(out_hw[0]), int(out_hw[1])
    d.ph_h, d.ph_w = int(phase_hw[0]), int(phase_hw[1])
    d.phd_h, d.phd_w = int(phase_dgrad_hw[0]), int(phase_dgrad_hw[1])
    if d.phd_h:                                        # x1 is the full-resolution dY: the launch's grid is the source resolution
        d.h = H // 2
    if d.phd_w:
        d.w_ = W // 2
    if w.shape[2] != d.c1 + d.c2:
        raise RhoHipError(f"conv: prepared weight has {w.shape[2]} input channels, inputs provide {d.c1 + d.c2}")
    if skip is not None:
        # (sx1, sx2, prepared 1x1x1 weights [1, coutp, c], bias): out += W_skip . cat(sx1, sx2) + b_skip inside this launch (rho_conv_desc.sk_*)
        sx1, sx2, sw, sb = skip
        d.sk_x1, d.sk_x2, d.sk_w, d.sk_bias = ptr(sx1), ptr(sx2), ptr(sw), ptr(sb)
        d.sk_c1 = sx1.shape[-1]
        d.sk_c2 = sx2.shape[-1] if sx2 is not None else 0
        if sw.shape[0] != 1 or sw.shape[1] != w.shape[1] or sw.shape[2] != d.sk_c1 + d.sk_c2:
            raise RhoHipError(f"conv: folded skip weights {tuple(sw.shape)} do not match coutp {w.shape[1]} / {d.sk_c1 + d.sk_c2} input channels")
    return d


def conv_out_shape(x_shape, kernel, stride_hw, up_hw):
    N, D, H, W, _ = x_shape
    kd, kh, kw = kernel
    ho = H * 2 if up_hw[0] else (H + 2 * (kh // 2) - kh) // stride_hw[0] + 1
    wo = W * 2 if up_hw[1] else (W + 2 * (kw // 2) - kw) // stride_hw[1] + 1
    return N, D, ho, wo


def conv_stats_tiles(desc: ConvDesc) -> int:
    """Tiles per sample of the fused output statistics of ``desc`` (0: not available for this geometry)."""
    return int(hip.lib().rho_conv_stats_tiles(C.byref(desc)))


def stem_conv3d_tiles(d: int, h: int, w: int) -> int:
    return int(hip.lib().rho_stem_conv3d_tiles(d, h, w))


def stem_conv3d(x: Tensor, w: Tensor, bias: Tensor, y: Tensor, stats: Optional[Tensor] = None) -> None:
    """rho_stem_conv3d: x float32 [N, 1, D, H, W] -> y bf16 channels-last [N, D, H, W, cout]; w = prepared im2col-form weights
    [1, coutp, 32]; stats (optional) float32 [N, stem_conv3d_tiles(D, H, W), 2, cout]."""
    hip.require_gpu(x, "x")
    N, _, D, H, W = x.shape
    check(hip.lib().rho_stem_conv3d(ptr(x), ptr(w), ptr(bias), ptr(y), ptr(stats), N, D, H, W, y.shape[-1], stream()), "rho_stem_conv3d")


def head_conv3d(x: Tensor, pre_a: Optional[Tensor], pre_b: Optional[Tensor], pre_silu: bool, w: Tensor, bias: Optional[Tensor],
                out: Tensor) -> None:
    """rho_head_conv3d: x bf16 channels-last [N, D, H, W, C] -> out float32 [N, 1, D, H, W]; w = prepared taps-as-rows weights [1, 32, C]."""
    hip.require_gpu(x, "x")
    N, D, H, W, Cc = x.shape
    check(hip.lib().rho_head_conv3d(ptr(x), ptr(pre_a), ptr(pre_b), int(pre_silu), ptr(w), ptr(bias), ptr(out), N, D, H, W, Cc, stream()),
          "rho_head_conv3d")


def conv_workspace_bytes(desc: ConvDesc) -> int:
    """Bytes of workspace the k-split of ``desc`` wants (rho_conv_desc.ws); 0: the launch is not split."""
    return int(hip.lib().rho_conv_workspace_bytes(C.byref(desc)))


def attach_conv_workspace(descs, device) -> Optional[Tensor]:
    """One workspace for all of ``descs`` (stream-ordered launches share it): the largest any of them wants, or None."""
    want = [conv_workspace_bytes(d) for d in descs]
    if not want or max(want) == 0:
        return None
    ws = torch.empty(max(want), dtype=torch.uint8, device=device)
    for d, b in zip(descs, want):
        if b:
            d.ws, d.ws_bytes = ptr(ws), ws.numel()
    return ws


def conv_variant(desc: ConvDesc) -> str:
    """Name of the k_conv instantiation rho_conv_nd_fwd launches for ``desc`` (nothing is launched)."""
    buf = C.create_string_buffer(128)
    check(hip.lib().rho_conv_variant(C.byref(desc), buf, 128), "rho_conv_variant")
    return buf.value.decode()


def conv_wgrad_variant(desc: ConvDesc, dy_width: int) -> str:
    buf = C.create_string_buffer(128)
    check(hip.lib().rho_conv_wgrad_variant(C.byref(desc), dy_width, buf, 128), "rho_conv_wgrad_variant")
    return buf.value.decode()


def conv_launch(desc: ConvDesc) -> None:
    check(hip.lib().rho_conv_nd_fwd(C.byref(desc), stream()), "rho_conv_nd_fwd")


def conv(x1: Tensor, x2: Optional[Tensor], w: Tensor, bias: Tensor, *, kernel, cout: int, split: Optional[int] = None,
         stride_hw=(1, 1), up_hw=(0, 0), pre_a=None, pre_b=None, pre_silu=False, res=None, res_add=None,
         res_add_stride: int = 0, y2_dtype: Optional[torch.dtype] = None):
    """Allocate outputs and run one convolution. Returns (y channels-last or None, y2 channel-major or None)."""
    split = cout if split is None else split
    N, Do, Ho, Wo = conv_out_shape(x1.shape, kernel, stride_hw, up_hw)
    y = torch.empty(N, Do, Ho, Wo, split, dtype=x1.dtype, device=x1.device) if split > 0 else None
    y2 = None
    if split < cout:
        y2 = torch.empty(N, cout - split, Do * Ho * Wo, dtype=y2_dtype or x1.dtype, device=x1.device)
    d = make_conv_desc(x1, x2, w, bias, kernel=kernel, cout=cout, split=split, y=y, y2=y2, stride_hw=stride_hw, up_hw=up_hw,
                       pre_a=pre_a, pre_b=pre_b, pre_silu=pre_silu, res=res, res_add=res_add, res_add_stride=res_add_stride)
    ws = attach_conv_workspace([d], x1.device)      # (freed to the caching allocator on return: reuse is ordered on this stream)
    conv_launch(d)
    return y, y2


# ----------------------------------------------------------------------------- attention
def attention(qk: Tensor, vt: Tensor, heads: int, out: Optional[Tensor] = None, lse: Optional[Tensor] = None) -> Tensor:
    """qk channels-last [B, T, 2C], vt channel-major [B, C, T] -> channels-last [B, T, C].
    lse (optional float32 [B, heads, T]) receives the per-query log-sum-exp for the backward."""
    B, T, C2 = qk.shape
    Cc = C2 // 2
    ch = Cc // heads
    out = torch.empty(B, T, Cc, dtype=qk.dtype, device=qk.device) if out is None else out
    check(hip.lib().rho_attention_fwd(ptr(qk), ptr(vt), ptr(out), ptr(lse), dtype_code(qk.dtype), B, T, heads, ch, stream()),
          "rho_attention_fwd")
    return out


# ============================================================================= backward ops
def prep_conv_weight_dgrad(w: Tensor, dtype: torch.dtype, col_src: Optional[Tensor] = None,
                           out: Optional[Tensor] = None) -> Tensor:
    """PyTorch conv weight [Cout, Cin, *k] -> dgrad weights [taps, ceil32(Cin), ceilCK(Cout)] (flipped, transposed)."""
    _f32c(w, "w")
    cout, cin = w.shape[0], w.shape[1]
    taps = int(math.prod(w.shape[2:])) if w.dim() > 2 else 1
    ck = elem_chunk(dtype)
    rowsp = ((cin + 31) // 32) * 32
    colsp = ((cout + ck - 1) // ck) * ck
    out = torch.empty(taps, rowsp, colsp, dtype=dtype, device=w.device) if out is None else out
    check(hip.lib().rho_prep_conv_weight_dgrad(ptr(w), ptr(out), dtype_code(dtype), cout, cin, taps, rowsp, colsp, ptr(col_src),
                                               stream()), "rho_prep_conv_weight_dgrad")
    return out


def conv_wgrad(desc: ConvDesc, dy: Tensor, dw: Tensor, dbias: Optional[Tensor] = None) -> None:
    """Accumulate the weight gradient of the forward conv `desc` into the fp32 buffer dw [taps, coutp, cin] (and, when given,
    the channel sums of dy = the bias gradient into dbias [coutp])."""
    if deterministic():
        need = int(hip.lib().rho_conv_wgrad_workspace_bytes(C.byref(desc), dy.shape[-1]))
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dy.device)
        check(hip.lib().rho_conv_nd_wgrad_ws(C.byref(desc), ptr(dy), dy.shape[-1], ptr(dw), ptr(dbias), ptr(ws), need, stream()),
              "rho_conv_nd_wgrad_ws")
        return
    check(hip.lib().rho_conv_nd_wgrad(C.byref(desc), ptr(dy), dy.shape[-1], ptr(dw), ptr(dbias), stream()), "rho_conv_nd_wgrad")


def wgrad_finalize(dw: Tensor, grad: Tensor, row_src: Optional[Tensor] = None, accumulate: bool = False) -> None:
    cout, cin = grad.shape[0], grad.shape[1]
    taps = grad.numel() // (cout * cin)
    check(hip.lib().rho_wgrad_finalize(ptr(dw), ptr(grad), cout, cin, taps, dw.shape[1], dw.shape[2], ptr(row_src), int(accumulate),
                                       stream()), "rho_wgrad_finalize")


def gn_bwd(g: Tensor, x1: Tensor, x2: Optional[Tensor], a: Tensor, b: Tensor, stats: Tensor, gamma: Tensor, beta: Tensor,
           pre_silu: bool, dx1: Tensor, dx2: Optional[Tensor], dgamma: Tensor, dbeta: Tensor, *, scale=None, film_stride=0,
           dscale=None, dshift=None, dfilm_stride=0, acc_params=False, acc1=False, acc2=False, ws=None, add1: Optional[Tensor] = None):
    """Backward of act(GN(x)*(1+scale)+shift); scale/dscale/dshift may be raw pointers (ints) or tensors.  ``add1``: a further
    addend of dx1 (a residual connection's gradient) folded into the apply pass."""
    N = x1.shape[0]
    c1 = x1.shape[-1]
    c2 = x2.shape[-1] if x2 is not None else 0
    Cc = c1 + c2
    S = x1.numel() // (N * c1)
    nblk = gn_nblk(S)
    dev = x1.device
    if ws is None:
        ws = dict(part=torch.empty(N * nblk * (Cc // 8) * 16, dtype=torch.float32, device=dev),
                  work=torch.empty(2 * N * Cc, dtype=torch.float32, device=dev),
                  cA=torch.empty(N, Cc, dtype=torch.float32, device=dev), cP=torch.empty(N, 32, dtype=torch.float32, device=dev),
                  cQ=torch.empty(N, 32, dtype=torch.float32, device=dev))
    L = hip.lib()
    dt = dtype_code(x1.dtype)
    p = lambda t: t if isinstance(t, int) or t is None else t.data_ptr()  # noqa: E731
    check(L.rho_gn_bwd_reduce(ptr(g), ptr(x1), c1, ptr(x2), c2, dt, N, S, ptr(a), ptr(b), ptr(stats), int(pre_silu), ptr(ws["part"]),
                              stream()), "rho_gn_bwd_reduce")
    check(L.rho_gn_bwd_finalize(ptr(ws["part"]), N, Cc, S, nblk, 0, ptr(gamma), ptr(beta), p(scale), film_stride, ptr(stats),
                                ptr(ws["work"]), ptr(dgamma), ptr(dbeta), int(acc_params), p(dscale), p(dshift), dfilm_stride,
                                ptr(ws["cA"]), ptr(ws["cP"]), ptr(ws["cQ"]), stream()), "rho_gn_bwd_finalize")
    check(L.rho_gn_bwd_apply(ptr(g), ptr(x1), c1, ptr(x2), c2, dt, N, S, ptr(a), ptr(b), int(pre_silu), ptr(ws["cA"]), ptr(ws["cP"]),
                             ptr(ws["cQ"]), ptr(dx1), ptr(dx2), int(acc1), int(acc2), ptr(add1), stream()), "rho_gn_bwd_apply")


def chan_sum(x: Tensor, out_nc: Tensor, out_c: Optional[Tensor] = None, nc_stride: int = 0, acc_nc=False, acc_c=False,
             partials: Optional[Tensor] = None) -> None:
    N, Cc = x.shape[0], x.shape[-1]
    S = x.numel() // (N * Cc)
    nblk = gn_nblk(S)
    partials = torch.empty(N * nblk * (Cc // 8) * 16, dtype=torch.float32, device=x.device) if partials is None else partials
    onc = out_nc if isinstance(out_nc, int) else out_nc.data_ptr()
    check(hip.lib().rho_chan_sum(ptr(x), dtype_code(x.dtype), N, S, Cc, ptr(partials), onc, nc_stride, int(acc_nc), ptr(out_c),
                                 int(acc_c), stream()), "rho_chan_sum")


def upsample2x(x: Tensor, up_hw, out: Optional[Tensor] = None) -> Tensor:
    N, D, H, W, Cc = x.shape
    out = torch.empty(N, D, H * (2 if up_hw[0] else 1), W * (2 if up_hw[1] else 1), Cc, dtype=x.dtype, device=x.device) if out is None else out
    check(hip.lib().rho_upsample2x(ptr(x), ptr(out), dtype_code(x.dtype), N * D, H, W, Cc, int(up_hw[0]), int(up_hw[1]), stream()),
          "rho_upsample2x")
    return out


def avgpool2x(x: Tensor, pool_hw, out: Optional[Tensor] = None) -> Tensor:
    """avg_pool_nd(kernel = stride = 2) over the flagged axes of a channels-last tensor (floor output extents)."""
    N, D, H, W, Cc = x.shape
    out = torch.empty(N, D, H // 2 if pool_hw[0] else H, W // 2 if pool_hw[1] else W, Cc, dtype=x.dtype, device=x.device) if out is None else out
    check(hip.lib().rho_avgpool2x(ptr(x), ptr(out), dtype_code(x.dtype), N * D, H, W, Cc, int(pool_hw[0]), int(pool_hw[1]), stream()),
          "rho_avgpool2x")
    return out


def avgpool2x_bwd(dy: Tensor, dx: Tensor, pool_hw, accumulate: bool = False) -> Tensor:
    N, D, H, W, Cc = dx.shape
    check(hip.lib().rho_avgpool2x_bwd(ptr(dy), ptr(dx), dtype_code(dx.dtype), N * D, H, W, Cc, int(pool_hw[0]), int(pool_hw[1]),
                                      int(accumulate), stream()), "rho_avgpool2x_bwd")
    return dx


def pool2x_sum(dy: Tensor, dx: Tensor, up_hw, accumulate: bool = False) -> Tensor:
    N, D, H, W, Cc = dx.shape
    check(hip.lib().rho_pool2x_sum(ptr(dy), ptr(dx), dtype_code(dx.dtype), N * D, H, W, Cc, int(up_hw[0]), int(up_hw[1]),
                                   int(accumulate), stream()), "rho_pool2x_sum")
    return dx


def linear_bwd(dout, x: Tensor, w: Tensor, dw: Optional[Tensor], db: Optional[Tensor], dx: Optional[Tensor],
               act_in: bool = False, acc_params: bool = False, acc_dx: bool = False, dout_stride: int = 0) -> None:
    """dout may be a tensor or a raw device pointer (a column slice of the batched FiLM gradient)."""
    B, K = x.shape
    O = w.shape[0]
    dp = dout if isinstance(dout, int) else dout.data_ptr()
    check(hip.lib().rho_linear_bwd(dp, dout_stride, ptr(x), ptr(w), ptr(dw), ptr(db), ptr(dx), B, K, O, int(act_in),
                                   int(acc_params), int(acc_dx), stream()), "rho_linear_bwd")


def add_inplace(dst: Tensor, src: Tensor) -> None:
    check(hip.lib().rho_add_inplace(ptr(dst), ptr(src), dtype_code(dst.dtype), dst.numel(), stream()), "rho_add_inplace")


def attention_bwd(qk: Tensor, vt: Tensor, o: Tensor, dout: Tensor, lse: Tensor, heads: int, dqkv: Optional[Tensor] = None,
                  delta_ws: Optional[Tensor] = None):
    """Returns dqkv channels-last [B, T, 3C] = (dq | dk | dv): the output gradient of the qkv projection."""
    B, T, C2 = qk.shape
    Cc = C2 // 2
    dqkv = torch.empty(B, T, 3 * Cc, dtype=qk.dtype, device=qk.device) if dqkv is None else dqkv
    delta_ws = torch.empty(B, heads, T, dtype=torch.float32, device=qk.device) if delta_ws is None else delta_ws
    esz = dqkv.element_size()
    check(hip.lib().rho_attention_bwd(ptr(qk), ptr(vt), ptr(o), ptr(dout), ptr(lse), ptr(delta_ws), dqkv.data_ptr(), 3 * Cc,
                                      dqkv.data_ptr() + 2 * Cc * esz, 3 * Cc, dtype_code(qk.dtype), B, T, heads, Cc // heads,
                                      stream()), "rho_attention_bwd")
    return dqkv
