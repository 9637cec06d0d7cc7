"""ctypes binding of librho_hip.so (C ABI: include/rho_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, the product path
raises.  PyTorch is used only for device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RHO_HIP_LIB: load an alternative build of the library (same-box A/B measurements) without touching the in-tree one
LIB_PATH = os.environ.get("RHO_HIP_LIB") or os.path.join(_HERE, "librho_hip.so")

RHO_F32 = 0
RHO_BF16 = 1
ABI_VERSION = 8      # == RHO_ABI_VERSION of include/rho_hip.h (tests/test_cabi.py keeps the two equal)

c_void_p, c_int, c_int32, c_int64, c_uint64, c_float, c_double = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double


class ConvDesc(C.Structure):
    """struct rho_conv_desc (include/rho_hip.h)."""
    _fields_ = [
        ("x1", c_void_p), ("x2", c_void_p), ("pre_a", c_void_p), ("pre_b", c_void_p), ("w", c_void_p),
        ("bias", c_void_p), ("res", c_void_p), ("res_add", c_void_p), ("y", c_void_p), ("y2", c_void_p),
        ("dtype", c_int32), ("y2_f32", c_int32), ("c1", c_int32), ("c2", c_int32),
        ("cout", c_int32), ("coutp", c_int32), ("split", c_int32),
        ("n", c_int32), ("d", c_int32), ("h", c_int32), ("w_", c_int32),
        ("kd", c_int32), ("kh", c_int32), ("kw", c_int32), ("sh", c_int32), ("sw", c_int32),
        ("up_h", c_int32), ("up_w", c_int32), ("pre_silu", c_int32), ("res_add_stride", c_int32),
        ("y2_cl", c_int32), ("zs_h", c_int32), ("zs_w", c_int32), ("out_h", c_int32), ("out_w", c_int32),
        ("res2", c_void_p), ("stats", c_void_p),
        ("ph_h", c_int32), ("ph_w", c_int32), ("phd_h", c_int32), ("phd_w", c_int32),
        ("gnb_x1", c_void_p), ("gnb_x2", c_void_p), ("gnb_c1", c_int32), ("gnb_silu", c_int32), ("gnb_a", c_void_p), ("gnb_b", c_void_p),
        ("ws", c_void_p), ("ws_bytes", c_int64),
        ("sk_x1", c_void_p), ("sk_x2", c_void_p), ("sk_w", c_void_p), ("sk_bias", c_void_p), ("sk_c1", c_int32), ("sk_c2", c_int32),
        ("gna_g", c_void_p), ("gna_cA", c_void_p), ("gna_cP", c_void_p), ("gna_cQ", c_void_p),
    ]


class PrepOp(C.Structure):
    """struct rho_prep_op (include/rho_hip.h): one prepared layout of one tensor in rho_prep_batch's device table."""
    _fields_ = [
        ("w", c_void_p), ("out", c_void_p), ("perm", c_void_p),
        ("cout", c_int64), ("cin", c_int64), ("d1", c_int64), ("d2", c_int64), ("total", c_int64),
        ("kind", c_int32), ("dtype", c_int32),
        ("kd", c_int32), ("kh", c_int32), ("kw", c_int32), ("kh2", c_int32), ("kw2", c_int32), ("ph_h", c_int32), ("ph_w", c_int32),
        ("sel_h", c_int32), ("sel_w", c_int32), ("flip_d", c_int32), ("dgrad", c_int32),
        ("blk0", c_int32), ("nblk", c_int32), ("pad_", c_int32),
    ]


PREP_FWD, PREP_DGRAD, PREP_PHASE, PREP_SEL, PREP_VEC = 0, 1, 2, 3, 4


class WfinOp(C.Structure):
    """struct rho_wfin_op (include/rho_hip.h): one accumulation buffer -> parameter gradient in rho_wgrad_finalize_batch's table."""
    _fields_ = [
        ("dw", c_void_p), ("grad", c_void_p), ("row_src", c_void_p),
        ("cout", c_int64), ("cin", c_int64), ("coutp", c_int64), ("cinb", c_int64), ("total", c_int64), ("phase_stride", c_int64),
        ("kind", c_int32), ("kd", c_int32), ("kh", c_int32), ("kw", c_int32), ("up_h", c_int32), ("up_w", c_int32),
        ("blk0", c_int32), ("nblk", c_int32),
    ]


# name -> (restype, argtypes); every symbol include/rho_hip.h declares
SIGNATURES = {
    "rho_abi_version": (c_int, []),
    "rho_build_info": (C.c_char_p, []),
    "rho_q_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "rho_p_sample_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rho_step_advance": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p]),
    "rho_philox_normal": (c_int, [c_void_p, c_int64, c_uint64, c_uint64, c_void_p, c_void_p]),
    "rho_mse": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rho_mse_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "rho_mean_flat": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "rho_set_deterministic": (c_int, [c_int]),
    "rho_get_deterministic": (c_int, []),
    "rho_adamw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float, c_int32, c_void_p]),
    "rho_timestep_embed": (c_int, [c_void_p] * 11 + [c_int64, c_int64, c_int64, c_int, c_void_p]),
    "rho_multi_embed": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "rho_multi_embed_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "rho_randint": (c_int, [c_void_p, c_int64, c_int64, c_uint64, c_uint64, c_void_p, c_void_p]),
    "rho_sph_harm_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "rho_sph_harm_fields": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rho_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "rho_pack_input": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "rho_prep_conv_weight": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "rho_prep_batch": (c_int, [c_void_p, c_int64, c_int64, c_void_p]),
    "rho_wgrad_finalize_batch": (c_int, [c_void_p, c_int64, c_int64, c_void_p]),
    "rho_prep_conv_weight_sel": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_int64, c_int64, c_int, c_void_p]),
    "rho_prep_conv_weight_phase": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64,
                                          c_int, c_void_p]),
    "rho_gn_nblk": (c_int, [c_int64]),
    "rho_gn_partial": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "rho_gn_finalize": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "rho_conv_nd_fwd": (c_int, [C.POINTER(ConvDesc), c_void_p]),
    "rho_conv_stats_tiles": (c_int64, [C.POINTER(ConvDesc)]),
    "rho_conv_workspace_bytes": (c_int64, [C.POINTER(ConvDesc)]),
    "rho_stem_conv3d_tiles": (c_int64, [c_int64, c_int64, c_int64]),
    "rho_stem_conv3d": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "rho_head_conv3d": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                c_int64, c_void_p]),
    "rho_conv_variant": (c_int, [C.POINTER(ConvDesc), C.c_char_p, c_int]),
    "rho_conv_wgrad_variant": (c_int, [C.POINTER(ConvDesc), c_int64, C.c_char_p, c_int]),
    "rho_gn_finalize2": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rho_im2col_taps": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int64, c_void_p]),
    "rho_tap_gather_sum": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int64, c_void_p, c_void_p,
                                   c_void_p]),
    "rho_attention_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    # ---- backward
    "rho_conv_nd_wgrad": (c_int, [C.POINTER(ConvDesc), c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "rho_conv_nd_wgrad_ws": (c_int, [C.POINTER(ConvDesc), c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rho_conv_wgrad_workspace_bytes": (c_int64, [C.POINTER(ConvDesc), c_int64]),
    "rho_wgrad_finalize": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int, c_void_p]),
    "rho_wgrad_finalize_phase": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int,
                                        c_void_p]),
    "rho_prep_conv_weight_dgrad": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "rho_gn_bwd_reduce": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p,
                                  c_void_p, c_int, c_void_p, c_void_p]),
    "rho_gn_bwd_finalize": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "rho_gn_bwd_apply": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rho_gn_apply_drop": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_float,
                                  c_uint64, c_void_p, c_void_p]),
    "rho_gn_bwd_reduce_drop": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                       c_int, c_void_p, c_float, c_uint64, c_void_p, c_void_p]),
    "rho_gn_bwd_apply_drop": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_uint64, c_void_p,
                                      c_void_p]),
    "rho_dropout_mask": (c_int, [c_void_p, c_int64, c_float, c_uint64, c_void_p, c_void_p]),
    "rho_gn_apply": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rho_ddpm_sched_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_float, c_float, c_float,
                                    c_float, c_float, c_void_p]),
    "rho_ema_update": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "rho_abs_quantile_workspace_bytes": (c_int64, [c_int64]),
    "rho_abs_quantile": (c_int, [c_void_p, c_int64, c_int64, c_double, c_void_p, c_void_p, c_void_p]),
    "rho_q_sample_coef": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                  c_void_p]),
    "rho_ddim_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_float,
                              c_float, c_float, c_float, c_void_p]),
    "rho_chan_sum": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "rho_upsample2x": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "rho_pool2x_sum": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "rho_avgpool2x": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "rho_avgpool2x_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "rho_linear_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int,
                               c_int, c_int, c_void_p]),
    "rho_add_inplace": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "rho_scale_by_device_scalar": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    # ---- legacy UNet (models/unet.py)
    "rho_act_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "rho_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "rho_groupnorm_act": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, c_float,
                                  c_int, c_void_p]),
    "rho_groupnorm_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                      c_int64, c_int64, c_int64, c_int, c_void_p]),
    "rho_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                  c_int, c_int64, c_int64, c_int64, c_int64, c_void_p]),
}


class RhoHipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the library and bind every declared symbol (works without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RhoHipError(
            f"{path} not found: build it with `python -m rho_diffusion_amd.build` (hipcc --offload-arch=gfx950). "
            "There is no CPU/PyTorch fallback for the product path.")
    lib = C.CDLL(path)
    check_abi(lib, path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check_abi(lib: C.CDLL, path: str = "") -> None:
    """Refuse a build whose rho_abi_version() differs from the signatures bound here (RHO_HIP_LIB / AB_REF point at arbitrary
    builds: one with every symbol but older argument lists would be called with shifted arguments)."""
    try:
        fn = lib.rho_abi_version
    except AttributeError as exc:
        raise RhoHipError(f"{path}: no rho_abi_version symbol - not a librho_hip build") from exc
    fn.restype, fn.argtypes = c_int, []
    v = int(fn())
    if v != ABI_VERSION:
        raise RhoHipError(f"{path}: ABI version {v}, this binding was written against {ABI_VERSION} (include/rho_hip.h); rebuild it")


def lib() -> C.CDLL:
    return load()


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = {-1: "RHO_E_ARG", -2: "RHO_E_ALIGN", -3: "RHO_E_SHAPE"}.get(rc, f"hipError {rc}")
        raise RhoHipError(f"{what} failed: {kind}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t: torch.Tensor, name: str = "tensor") -> None:
    if not t.is_cuda:
        raise RhoHipError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return RHO_F32
    if dt == torch.bfloat16:
        return RHO_BF16
    raise RhoHipError(f"unsupported engine dtype {dt}")
