"""CPU: the C-ABI library loads without a GPU and exports every symbol include/rho_hip.h declares,
and the ctypes binding declares the same set (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rho_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rho_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ["rho_q_sample", "rho_p_sample_step", "rho_philox_normal", "rho_conv_nd_fwd", "rho_attention_fwd",
                 "rho_gn_partial", "rho_gn_finalize", "rho_adamw", "rho_mse", "rho_linear", "rho_timestep_embed", "rho_multi_embed",
                 "rho_randint", "rho_sph_harm_fields", "rho_conv_variant"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    from rho_diffusion_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        from rho_diffusion_amd.build import build
        build(verbose=False)
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in rho_hip.h but not exported"


def test_ctypes_binding_matches_header():
    from rho_diffusion_amd import hip
    assert sorted(hip.SIGNATURES.keys()) == _declared()
    lib = hip.load()
    header = open(os.path.join(ROOT, "include", "rho_hip.h")).read()
    declared = int(re.search(r"#define\s+RHO_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.rho_abi_version() == declared == hip.ABI_VERSION
    assert b"gfx950" in lib.rho_build_info()


def test_loader_refuses_a_build_with_another_abi_version():
    """hip.check_abi (used by hip.load and the tools/ab_*.py probes): a library reporting another version must not be bound."""
    from rho_diffusion_amd import hip

    class _Fn:
        restype = argtypes = None

        def __init__(self, v):
            self.v = v

        def __call__(self):
            return self.v

    class _Lib:
        def __init__(self, v):
            self.rho_abi_version = _Fn(v)

    hip.check_abi(_Lib(hip.ABI_VERSION), "same")
    with pytest.raises(hip.RhoHipError, match="ABI version"):
        hip.check_abi(_Lib(hip.ABI_VERSION - 1), "older build")
    with pytest.raises(hip.RhoHipError, match="rho_abi_version"):
        hip.check_abi(object(), "not ours")


def test_conv_desc_struct_layout():
    """ctypes struct must mirror `struct rho_conv_desc` field for field."""
    from rho_diffusion_amd.hip import ConvDesc
    text = open(os.path.join(ROOT, "include", "rho_hip.h")).read()
    body = text[text.index("typedef struct rho_conv_desc {"):text.index("} rho_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(",")
        first = names[0].split()[-1].lstrip("*")
        fields.append(first)
        fields.extend(n.strip().lstrip("*") for n in names[1:])
    assert [f[0] for f in ConvDesc._fields_] == fields
    assert ctypes.sizeof(ConvDesc) == 10 * 8 + 26 * 4 + 8 + 8 + 4 * 4 + (2 * 8 + 2 * 4 + 2 * 8) + 2 * 8 + (4 * 8 + 2 * 4)
