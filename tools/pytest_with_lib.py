#!/usr/bin/env python
"""Run pytest against another build of the HIP library (development tool: parity of an A/B variant from tools/build_variant.py).
Usage: python tools/pytest_with_lib.py storygen_amd/lib/libstorygen_hip_<suffix>.so [pytest arguments...]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from storygen_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest  # noqa: E402

sys.exit(pytest.main(sys.argv[2:]))
