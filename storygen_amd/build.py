"""Builds libstorygen_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m storygen_amd.build [--force] [--experiments]

--experiments (or SG_BUILD_EXPERIMENTS=1) adds the instrumented kernel instantiations behind sg_debug_gemm_anatomy /
sg_debug_conv_anatomy (tools/anatomy.py); the product library is built without them.

hipcc cross-compiles for gfx950 without a GPU.  The shared object lands in storygen_amd/lib/ (git-ignored,
but shipped to the GPU box by gpurun) and is what storygen_amd/_lib.py dlopens.
"""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libstorygen_hip.so")
LIB_EXP = os.path.join(LIBDIR, "libstorygen_hip_exp.so")      # --experiments: a SEPARATE library that only tools/anatomy.py loads
SOURCES = ["gemm_conv.hip", "attention.hip", "attention_f8.hip", "norm.hip", "misc.hip", "backward.hip", "attention_bwd.hip", "encoders.hip", "optim.hip", "ff_fused.hip"]
# -fno-slp-vectorize for EVERY kernel (round 6): no packed fp32 VALU (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel).
#  * correctness: the one run-to-run difference this project ever saw — the columns-are-tokens LayerNorm fold on the 4-stage latency
#    kernel under the two-branch graph — was ONE half of a `v_pk_add_f32 ..., v[c:d] op_sel:[0,1]` in the epilogue losing its d term
#    for lanes 48 - 63 (the output was exactly `right - d_row` on 16 columns of a row; inputs bit-identical; profiles/r06bm_*).  With
#    the same source built without SLP packing: 40 / 40 and 40 / 40 repeats bit-identical where the packed build gave 34 / 40 and
#    1 / 40 (profiles/r06bn_*).  Two workgroups of that kernel share a CU; whatever the hardware condition is, the compiler's hazard
#    recogniser does not know it, so nothing in this library issues packed fp32 arithmetic.
#  * speed: neutral on the sampler's step (gemm_conv.hip / attention.hip: 12.39 / 12.37 vs 12.41 / 12.43 ms), -30 % on the backward
#    attention's dK/dV pass, which ran such instructions beside MFMAs (the microarchitecture guide's anti-lever).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize",
         "-Rpass-analysis=kernel-resource-usage"]          # remarks only: parsed into lib/kernel_resources.json
RESOURCES = os.path.join(LIBDIR, "kernel_resources.json")
# attention keeps its O^T accumulators live across the softmax VALU code of every tile: with MFMA results in AGPRs the
# compiler shuttles them through v_accvgpr_read/write around each tile (137 of 281 VALU instructions per tile,
# profiles/r01d_pmc_kernels.txt); the VGPR form of MFMA (gfx950's register file is unified) removes all of them.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "attention_f8.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               # (round 6) the backward kernel turns S^T / dP^T into P / dS with VALU code between two groups of MFMAs: same reason
               # ... and no SLP packing: v_pk_add_f32 / v_pk_mul_f32 in the P / dS code beside the MFMAs cost the dK/dV pass 30 %
               # (417 -> 293 us at B4 Nq 4096 Nk 4096, profiles/r06bc_*; the microarchitecture guide's "packed fp32 VALU is an anti-lever")
               "attention_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               # the GELU polynomial of the fused feed-forward runs beside MFMAs: packed fp32 VALU (what SLP vectorisation makes of it)
               # is slower there than the scalar forms (MI355X_MICROARCH.md, price of fillers beside MFMAs)
               "ff_fused.hip": []}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


_REMARK = re.compile(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                     r"LDS Size \[bytes/block\]): +(\S+)")
_KEYS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
         "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "LDS Size [bytes/block]": "lds_bytes_per_block"}


def _parse_resource_remarks(out: str, src: str, into: dict) -> str:
    """Moves hipcc's -Rpass-analysis=kernel-resource-usage remarks of one translation unit into `into`
    ({mangled kernel name: {...}}); returns the remaining compiler output (real warnings)."""
    rest, cur, in_remark, pending = [], None, False, []
    for line in out.splitlines():
        if line.startswith("In file included from"):        # include stack: belongs to whatever diagnostic follows
            pending.append(line)
            continue
        if "remark:" not in line:
            rest.extend(pending)
        pending = []
        m = _REMARK.search(line)
        if re.match(r"^\s*\d*\s*\|", line) and in_remark:     # source snippet under a remark
            continue
        in_remark = "remark:" in line
        if m:
            if m.group(1) == "Function Name":
                cur = into.setdefault(m.group(2), {"source": src})
            elif cur is not None:
                cur[_KEYS[m.group(1)]] = int(m.group(2))
        elif "-Rpass-analysis=kernel-resource-usage" not in line and "argument unused during compilation" not in line:
            rest.append(line)
    return "\n".join(rest)


def source_hash() -> str:
    """sha256 over the kernel sources and the C ABI header (16 hex digits): stamped into profiles/traffic.json by
    tools/traffic_from_pmc.py so that bench.py can tell whether the PMC traffic on file was measured on THIS build."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "storygen_hip.h")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(RESOURCES):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG, "..", "include", "storygen_hip.h"),
                                                                os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
    experiments = experiments or os.environ.get("SG_BUILD_EXPERIMENTS") == "1"
    if not force and not experiments and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    exp_flags = ["-DSG_BUILD_EXPERIMENTS=1"] if experiments else []
    objdir = os.path.join(LIBDIR, "exp") if experiments else LIBDIR
    os.makedirs(objdir, exist_ok=True)
    out_lib = LIB_EXP if experiments else LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *exp_flags, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    resources = {}
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        rest = _parse_resource_remarks(out, src, resources)
        if verbose and rest.strip():
            print(rest)
    with open(RESOURCES + (".exp" if experiments else ""), "w") as f:
        json.dump(resources, f, indent=1, sort_keys=True)
    # a kernel with a private segment either spills or keeps an array in memory: both are order-of-magnitude cliffs on this
    # hardware (the D = 160 attention instantiation once ran 6x slower that way) — refuse to ship one silently
    # (mma_fat_kernel — eight waves of 128x64, a measured negative reachable by tile hint only — has 256 registers per wave for 128
    # accumulators and spills 30 - 60 dwords in its EPILOGUE, none in the mainloop: tolerated, listed, checked to stay small)
    bad = {k: v["scratch_bytes_per_lane"] for k, v in resources.items()
           if v["scratch_bytes_per_lane"] and not ("mma_fat_kernel" in k and v["scratch_bytes_per_lane"] <= 256)}
    if bad and os.environ.get("SG_ALLOW_SCRATCH") != "1":
        raise RuntimeError(f"kernels using scratch memory (set SG_ALLOW_SCRATCH=1 to build anyway): {bad}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out_lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out_lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments="--experiments" in sys.argv))
