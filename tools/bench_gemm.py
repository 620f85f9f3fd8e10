#!/usr/bin/env python
"""Tile-shape sweep of sg_gemm_f16 / sg_conv3x3 on the layer shapes of BASELINE config 2 (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402

ops.apply_env_options()      # SG_* development variables -> sg_debug_set_option

dev = torch.device("cuda:0")
TILES = [(0, 0, False), (256, 128, False), (128, 128, False), (256, 64, False), (128, 64, False), (64, 128, False), (64, 64, False), (128, 128, True)]
FAT = []


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(40_000_000)   # ~20 ms of GPU spin: lets the host enqueue everything, so launches run back-to-back
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


def main():
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    gemms = [(12288, 320, 320), (12288, 320, 2880), (12288, 960, 320), (12288, 2560, 320), (12288, 320, 1280), (3072, 640, 640),
             (3072, 640, 5760), (768, 1280, 1280), (768, 1280, 11520), (192, 1280, 11520), (192, 1280, 1280), (36864, 640, 320)]
    print("GEMM  (us | TFLOP/s) per tile; last columns = generic kernel")
    print(f"{'shape':24s}" + "".join(f"{f'{t[0]}x{t[1]}' + ('g' if t[2] else ''):>16s}" for t in TILES)
          + "".join(f"{f'{t[0]}x{t[1]}/{t[2]}w':>16s}" for t in FAT))
    for M, N, K in gemms:
        a = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        out = torch.empty(M, N, dtype=torch.float16, device=dev)
        row = f"{f'M{M} N{N} K{K}':24s}"
        for t in TILES:
            ops.debug_set_tile(*t)
            us = timeit(lambda: ops.gemm(a, w, out, workspace=ws))
            row += f"{us:8.1f}|{2.0 * M * N * K / us / 1e6:6.0f} "
        ops.debug_set_tile(0, 0, False)
        for t in FAT:
            us = timeit(lambda: ops.gemm(a, w, out, workspace=ws, tile=t))
            row += f"{us:8.1f}|{2.0 * M * N * K / us / 1e6:6.0f} "
        print(row, flush=True)
    convs = [(4, 64, 64, 320, 320), (4, 32, 32, 640, 640), (4, 16, 16, 1280, 1280), (4, 8, 8, 1280, 1280), (4, 64, 64, 640, 320),
             (3, 64, 64, 320, 320), (3, 16, 16, 2560, 1280)]
    print("CONV3x3 padded input")
    for B, H, W, Ci, Co in convs:
        xp = torch.zeros(B, H + 2, W + 2, Ci, dtype=torch.float16, device=dev)
        xp[:, 1:-1, 1:-1] = torch.randn(B, H, W, Ci, device=dev).half()
        w = (torch.randn(Co, 3, 3, Ci, device=dev) / (9 * Ci) ** 0.5).half()
        out = torch.empty(B, H, W, Co, dtype=torch.float16, device=dev)
        row = f"{f'B{B} {H}x{W} {Ci}->{Co}':24s}"
        for t in TILES:
            ops.debug_set_tile(*t)
            us = timeit(lambda: ops.conv3x3(xp, w, out, workspace=ws, x_padded=True))
            row += f"{us:8.1f}|{2.0 * B * H * W * Co * 9 * Ci / us / 1e6:6.0f} "
        ops.debug_set_tile(0, 0, False)
        for t in FAT:
            us = timeit(lambda: ops.conv3x3(xp, w, out, workspace=ws, x_padded=True, tile=t))
            row += f"{us:8.1f}|{2.0 * B * H * W * Co * 9 * Ci / us / 1e6:6.0f} "
        print(row, flush=True)
    ops.debug_set_tile(0, 0, False)


if __name__ == "__main__":
    main()
