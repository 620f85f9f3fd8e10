#!/usr/bin/env python
"""Run bench.py against another build of the HIP library (same-box A/B of two kernel-source states; development tool).
Usage: python tools/ab_lib.py storygen_amd/lib/libstorygen_hip_s1.so [bench.py arguments...]"""
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from storygen_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
