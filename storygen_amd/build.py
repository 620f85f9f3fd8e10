"""Builds libstorygen_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m storygen_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The shared object lands in storygen_amd/lib/ (git-ignored,
but shipped to the GPU box by gpurun) and is what storygen_amd/_lib.py dlopens.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libstorygen_hip.so")
SOURCES = ["gemm_conv.hip", "attention.hip", "norm.hip", "misc.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# attention keeps its O^T accumulators live across the softmax VALU code of every tile: with MFMA results in AGPRs the
# compiler shuttles them through v_accvgpr_read/write around each tile (137 of 281 VALU instructions per tile,
# profiles/r01d_pmc_kernels.txt); the VGPR form of MFMA (gfx950's register file is unified) removes all of them.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG, "..", "include", "storygen_hip.h"),
                                                                os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
