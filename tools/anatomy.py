#!/usr/bin/env python
"""Where the cycles of the pipelined GEMM / conv mainloop go (development tool, round 2).

Launches each shape through the instrumented instantiation (sg_debug_gemm_anatomy / sg_debug_conv_anatomy: s_memtime stamps
around the phases of every 64-deep slab, summed per wave) and prints, per tile shape, the mean cycles per slab spent in
  wait   : s_waitcnt vmcnt(N)  (this wave's LDS-DMA of the slab has landed)
  barrier: s_barrier           (every wave's has)
  head   : first fragment ds_reads + the 4 MFMAs of k-step 0 (exposes LDS latency)
  pre    : k-step 1's fragment prefetch up to the DMA issue
  issue  : the global_load_lds instructions of the slab two ahead
  rest   : remaining 12 MFMAs + fragment reads
plus prologue / epilogue / total per wave.  MFMA issue alone would be 16 x 32 = 512 cycles per slab per wave (x2 when two waves
share a SIMD).  Usage: python tools/anatomy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import _lib, build  # noqa: E402

if not os.path.exists(build.LIB_EXP):
    sys.exit("tools/anatomy.py needs the experiments library: python -m storygen_amd.build --experiments")
_lib.LIB_PATH = build.LIB_EXP       # the instrumented instantiations live in a separate library (never the product's)
from storygen_amd import ops  # noqa: E402

ops.apply_env_options()      # SG_* development variables -> sg_debug_set_option

dev = torch.device("cuda:0")
NAMES = ["slabs", "wait", "barrier", "head", "pre", "issue", "rest", "prologue", "epilogue", "total"]


def report(name, tile, nwaves, run):
    prof = torch.zeros(10 * 400_000, dtype=torch.int64, device=dev)
    run(None)                       # warm (normal kernel)
    torch.cuda.synchronize()
    ops.ANATOMY = prof
    try:
        run(tile)
        torch.cuda.synchronize()
    finally:
        ops.ANATOMY = None
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run(tile)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    rows = prof.view(-1, 10)
    rows = rows[rows[:, 0] > 0].double()
    if rows.numel() == 0:
        print(f"{name:30s} {tile}: no profile rows (not the pipelined kernel?)")
        return
    slabs = rows[:, 0].mean()
    per = rows[:, 1:7].sum(0) / rows[:, 0].sum()
    pro, epi, tot = rows[:, 7].mean(), rows[:, 8].mean(), rows[:, 9].mean()
    print(f"{name:30s} {tile[0]:3d}x{tile[1]:<3d} {us:6.1f} us  waves {rows.shape[0]:5d} slabs/wave {slabs:5.1f} | per slab: "
          + " ".join(f"{n} {v:6.0f}" for n, v in zip(NAMES[1:7], per.tolist()))
          + f" = {per.sum():6.0f} | prologue {pro:6.0f} epilogue {epi:6.0f} total {tot:7.0f} cyc ({tot / 100:6.1f} us @100MHz ticks?)", flush=True)


def main():
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    print(f"development options: {ops.apply_env_options()}")
    for (B, H, Ci, Co) in [(3, 64, 320, 320), (3, 16, 1280, 1280)]:
        xp = torch.zeros(B, H + 2, H + 2, Ci, dtype=torch.float16, device=dev)
        xp[:, 1:-1, 1:-1] = torch.randn(B, H, H, Ci, device=dev).half()
        w = (torch.randn(Co, 3, 3, Ci, device=dev) / (9 * Ci) ** 0.5).half()
        out = torch.empty(B, H, H, Co, dtype=torch.float32, device=dev)
        res = torch.randn(B, H, H, Co, device=dev)
        for tile in [(256, 128), (256, 64), (128, 128), (128, 64), (64, 64)]:
            report(f"conv B{B} {H}x{H} {Ci}->{Co}", tile, 0,
                   lambda t: ops.conv3x3(xp, w, out, res1=res, split_k=1, workspace=ws, x_padded=True, tile=t))
    for (M, N, K) in [(12288, 320, 320), (768, 1280, 1280), (3072, 5120, 640)]:
        a = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        for tile in [(256, 128), (256, 64), (128, 128), (128, 64), (64, 64)]:
            report(f"gemm M{M} N{N} K{K}", tile, 0, lambda t: ops.gemm(a, w, out, split_k=1, workspace=ws, tile=t))
    # the residual-stream projections (attention.py:262,277): fp32 output + fp32 residual + bias
    for (M, N, K) in [(12288, 320, 320), (16384, 320, 320), (3072, 640, 640)]:
        a = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        out, res, bias = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev), torch.randn(N, device=dev).half()
        for tile in [(256, 64), (128, 128), (128, 64), (64, 64)]:
            report(f"gemm+res32 M{M} N{N} K{K}", tile, 0, lambda t: ops.gemm(a, w, out, bias=bias, res1=res, split_k=1, workspace=ws, tile=t))


if __name__ == "__main__":
    main()
