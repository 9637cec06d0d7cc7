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
    assert ctypes.sizeof(ConvDesc) == 10 * 8 + 26 * 4 + 8 + 8 + 4 * 4 + (2 * 8 + 2 * 4 + 2 * 8) + 2 * 8 + (4 * 8 + 2 * 4) + 4 * 8


def _gfx950_code_objects(path):
    """(offset, size) of every gfx950 code object in the library's clang offload bundles."""
    import struct
    blob = open(path, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", blob):
        p = m.start()
        (n,) = struct.unpack_from("<Q", blob, p + 24)
        o = p + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, o)
            o += 24
            triple = blob[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                out.append(blob[p + off:p + off + size])
    return out


def test_m0_is_only_written_by_the_lds_dma_statements(tmp_path):
    """k_wgrad's LDS-DMA statements leave M0 (the DMA's LDS base) as they set it instead of saving and restoring it around
    every DMA; that is only sound while nothing the compiler emits reads or writes M0.  Disassemble the built library:
    every instruction that mentions m0 must be a scalar move to / from m0, and every move TO m0 must be followed, within
    two instructions, by the global_load_lds it serves."""
    import shutil
    import subprocess
    from rho_diffusion_amd import hip
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump) or not os.path.exists(hip.LIB_PATH):
        pytest.skip("no llvm-objdump / library")
    objs = _gfx950_code_objects(hip.LIB_PATH)
    assert objs, "no gfx950 code object in the library"
    n_dma = 0
    for i, blob in enumerate(objs):
        f = tmp_path / f"co{i}.o"
        f.write_bytes(blob)
        text = subprocess.run([objdump, "-d", "--no-show-raw-insn", str(f)], capture_output=True, text=True, check=True).stdout
        if "m0" not in text:
            continue
        lines = [l.strip() for l in text.split("\n")]
        for k, l in enumerate(lines):
            if not re.search(r"\bm0\b", l):
                continue
            assert re.match(r"s_mov_b32 (m0, s\d+|s\d+, m0)\b", l), f"unexpected use of m0: {l}"
            if l.startswith("s_mov_b32 m0"):
                nxt = [x for x in lines[k + 1:k + 4] if x]
                assert any(x.startswith("global_load_lds") for x in nxt[:2]), f"m0 written without a DMA behind it: {l} / {nxt}"
                n_dma += 1
    assert n_dma > 0
    shutil.rmtree(tmp_path, ignore_errors=True)
