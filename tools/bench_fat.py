#!/usr/bin/env python
"""The batch-20 convolutions of the reference pass on the shipped plan vs the 128x64-per-wave tiles (mma_fat_kernel: tile hints (512, 128, 8) /
(256, 256, 8)), back to back, event-timed (development tool).
Usage: python tools/bench_fat.py [reps=20]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402

ops.apply_env_options()
dev = torch.device("cuda:0")
reps = next((int(a[5:]) for a in sys.argv[1:] if a.startswith("reps=")), 20)
B = 20
# (H, W, Cin, Cout, upsample): the stride-1 convolutions of one batched reference pass
SHAPES = [(64, 64, 320, 320, False), (64, 64, 640, 320, False), (64, 64, 960, 320, False), (32, 32, 320, 640, False), (32, 32, 640, 640, False),
          (32, 32, 1280, 640, False), (32, 32, 1920, 640, False), (16, 16, 640, 1280, False), (16, 16, 1280, 1280, False), (16, 16, 2560, 1280, False),
          (16, 16, 1280, 1280, True), (32, 32, 640, 640, True), (8, 8, 1280, 1280, False)]


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print(f"{'conv3x3 B20':34s} {'plan':>22s} {'us':>9s} {'TFLOP/s':>8s} {'512x128 us':>11s} {'TFLOP/s':>8s} {'256x256 us':>11s} {'TFLOP/s':>8s}")
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
for H, W, Ci, Co, ups in SHAPES:
    Hi, Wi = (H // 2, W // 2) if ups else (H, W)
    xp = torch.zeros(B, Hi + 2, Wi + 2, Ci, dtype=torch.float16, device=dev)
    xp[:, 1:-1, 1:-1] = torch.randn(B, Hi, Wi, Ci, device=dev).half()
    w = (torch.randn(Co, 3, 3, Ci, device=dev) * (9 * Ci) ** -0.5).half()
    out = torch.empty(B, H, W, Co, dtype=torch.float32, device=dev)
    bias, rb = torch.randn(Co, device=dev).half(), torch.randn(B, Co, device=dev)
    flops = 2.0 * B * H * W * Co * 9 * Ci
    row = f"{H}x{W} {Ci}->{Co}{' up' if ups else ''}".ljust(34)
    for tile in (None, (512, 128, 8), (256, 256, 8)):
        kw = dict(upsample2x=ups, bias=bias, rowbias=rb, x_padded=True, workspace=ws, tile=tile)
        hw = H * W
        bm = 0
        try:
            d, _, _ = ops._conv_desc(xp, w, out, **kw)
            plan = ops._plan_of(ops.lib.sg_conv3x3_launch_plan, d)
            bm = plan[0]
            if tile is not None and plan[5] != 2:
                raise ValueError("not applicable")
            if hw % bm == 0:
                kw["stats"] = torch.zeros(B * hw // bm * 2 * Co, dtype=torch.float32, device=dev)
                if ops.conv3x3_stats_rows(xp, w, out, **kw) != bm:
                    kw.pop("stats")
            t = timed(lambda: ops.conv3x3(xp, w, out, **kw))
            if tile is None:
                row += f" {plan[0]}x{plan[1]} s{plan[2]} {'stats' if 'stats' in kw else ''}".rjust(23)
            row += f" {t:9.1f} {flops / t * 1e-6:8.1f}" if tile is None else f" {t:11.1f} {flops / t * 1e-6:8.1f}"
        except Exception as e:        # a tile that does not apply to the shape
            row += f" {'-':>11s} {'-':>8s}"
    print(row, flush=True)
