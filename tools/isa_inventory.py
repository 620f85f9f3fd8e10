#!/usr/bin/env python
"""Static instruction inventory of a kernel's loops (development tool; CPU only: hipcc cross-compiles gfx950).
Compiles one csrc/*.hip to assembly, finds the named kernel, and prints for every loop (backward branch) the instruction mix —
MFMA / transcendental / other VALU / LDS / vector memory / SALU — and the VALU opcode histogram of loops that contain MFMAs.
Usage: python tools/isa_inventory.py attention.hip attn_fwd_kernelILi40ELi4ELi3ELi1ELb0ELb0E [extra hipcc flags...]
(attention sources need `-mllvm -amdgpu-mfma-vgpr-form=1`, as in storygen_amd/build.py)"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    return "salu" if op.startswith("s_") else "other"


def main():
    src, needle, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only",
                               f"-I{ROOT}/include", "-o", out, os.path.join(ROOT, "storygen_amd", "csrc", src)] + extra, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if needle in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\S+:", l)]
    if not starts:
        sys.exit(f"no kernel symbol containing {needle!r}")
    start = starts[0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    print(lines[start].split(":")[0])
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = set()
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.add((labels[m.group(1)], i, m.group(1)))
    for a, b, t in sorted(loops):
        mix, ops = Counter(), Counter()
        for l in body[a:b + 1]:
            s = l.strip()
            if not s or s[0] in ";." or s.endswith(":"):
                continue
            op = s.split()[0]
            mix[classify(op)] += 1
            ops[op] += 1
        print(f"loop {t} ({b - a} lines): {dict(mix)}")
        if mix["mfma"] >= 4:
            print("   VALU opcodes:", ", ".join(f"{k} x{v}" for k, v in ops.most_common() if k.startswith("v_") and not k.startswith("v_mfma")))


if __name__ == "__main__":
    main()
