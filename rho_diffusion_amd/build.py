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
           "sph_harm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-I", INCLUDE, "-I", CSRC]


def _newer(src_paths, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_paths)


def build(verbose: bool = True, force: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "rho_hip.h")]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            jobs.append([hipcc, *FLAGS, "-c", src, "-o", obj])

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
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
