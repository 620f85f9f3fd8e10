#!/usr/bin/env python
"""A/B build of the HIP library with ONE translation unit recompiled under extra flags (development tool; CPU only).
Usage: python tools/build_variant.py <suffix> <file.hip> [extra hipcc flags...]
       -> storygen_amd/lib/libstorygen_hip_<suffix>.so = the current objects of the other sources + this file rebuilt.
Load it with tools/ab_lib.py / the library argument of tools/bench_attn_bwd.py.  (Variants may use scratch: the shipped build refuses it.)"""
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from storygen_amd import build as B  # noqa: E402

suffix, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build(verbose=False)                                   # the objects of the other sources must be current
obj = os.path.join(B.LIBDIR, f"{src[:-4]}_{suffix}.o")
flags = [f for f in B.FLAGS if not f.startswith("-Rpass")]
subprocess.check_call([B._hipcc(), *flags, *B.EXTRA_FLAGS.get(src, []), *extra, "-c", os.path.join(B.CSRC, src), "-o", obj])
objs = [obj if s == src else os.path.join(B.LIBDIR, s.replace(".hip", ".o")) for s in B.SOURCES]
out = os.path.join(B.LIBDIR, f"libstorygen_hip_{suffix}.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
print(out)
