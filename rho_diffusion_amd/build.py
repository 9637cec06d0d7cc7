"""Build librho_hip.so (the C-ABI library of include/rho_hip.h) with hipcc for gfx950.

In-tree, explicit ``hipcc -shared -fPIC``: the built .so travels with the source snapshot to the
GPU box (it is git-ignored, not gpurun-ignored).  Objects are rebuilt only when their source or
a header is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "librho_hip.so")
SOURCES = ["elementwise.hip", "groupnorm.hip", "conv.hip", "wgrad.hip", "attention.hip", "attention_bwd.hip", "select.hip", "embedding.hip",
           "sph_harm.hip", "unet_v1.hip", "ends.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I", INCLUDE, "-I", CSRC]


def _newer(src_paths, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_paths)


def source_id() -> str:
    """sha256 (12 hex digits) over every source and header of the library: reported by rho_build_info(), stored next to
    profiler summaries, so a measurement can be matched to the binary it was taken on."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [os.path.join(INCLUDE, "rho_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def build(verbose: bool = True, force: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "rho_hip.h")]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    sid = source_id()
    idfile = os.path.join(CSRC, ".build_id")
    old = open(idfile).read().strip() if os.path.exists(idfile) else ""
    objs = []
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        stamp = s == "elementwise.hip" and old != sid          # rho_build_info() lives there and carries the id
        if force or stamp or _newer([src] + headers, obj):
            jobs.append([hipcc, *FLAGS, f'-DRHO_BUILD_ID="{sid}"', "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(objs, LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    with open(idfile, "w") as f:
        f.write(sid)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
