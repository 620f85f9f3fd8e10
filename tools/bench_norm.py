#!/usr/bin/env python
"""Isolated timings of the bandwidth-bound kernels of one denoising step (GroupNorm at every shape the SD-1.5 UNet
uses at a 64x64 latent, LayerNorm), with the bytes each launch has to move.  Development tool; the kernel variant is
chosen by the SG_GN_* environment knobs of the process (see norm.hip), so A/B = two runs of this script."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402

ops.apply_env_options()      # SG_* development variables -> sg_debug_set_option

GN_SHAPES = [(64, 320), (64, 640), (64, 960), (32, 320), (32, 640), (32, 960), (32, 1280), (32, 1920),
             (16, 640), (16, 1280), (16, 1920), (16, 2560), (8, 1280), (8, 2560)]
LN_SHAPES = [(4096, 320), (1024, 640), (256, 1280)]


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    torch.cuda._sleep(20_000_000)   # ~10 ms of GPU spin: the host enqueues everything meanwhile, launches run back-to-back
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    dev = "cuda:0"
    knobs = {k: v for k, v in os.environ.items() if k.startswith("SG_")}
    print(f"knobs: {knobs}")
    print(f"{'op':10s} {'B':>2s} {'side':>4s} {'C':>5s} {'MB':>7s} {'us':>8s} {'GB/s':>8s}")
    for B in (4, 3):
        for side, C in GN_SHAPES:
            hw = side * side
            x = torch.randn(B, hw, C, device=dev)
            g, b = torch.randn(C, device=dev).half(), torch.randn(C, device=dev).half()
            y = torch.zeros(B, side + 2, side + 2, C, dtype=torch.float16, device=dev)
            ws = torch.empty(ops.groupnorm_workspace_bytes(B, 32), dtype=torch.uint8, device=dev)
            us = timed(lambda: ops.groupnorm(x, g, b, y, 32, 1e-5, True, ws))
            mb = B * hw * C * 6 / 1e6       # one fp32 read + one fp16 write
            print(f"{'groupnorm':10s} {B:2d} {side:4d} {C:5d} {mb:7.2f} {us:8.2f} {mb / us * 1e3:8.0f}")
    for B in (4, 3):
        for hw, C in LN_SHAPES:
            M = B * hw
            x = torch.randn(M, C, device=dev)
            g, b = torch.randn(C, device=dev).half(), torch.randn(C, device=dev).half()
            y1, y2 = (torch.empty(M, C, dtype=torch.float16, device=dev) for _ in range(2))
            us = timed(lambda: ops.layernorm(x, g, b, y1, 1e-5, g, b, y2))
            mb = M * C * 8 / 1e6
            print(f"{'layernorm2':10s} {B:2d} {hw:4d} {C:5d} {mb:7.2f} {us:8.2f} {mb / us * 1e3:8.0f}")


def groupnorm_with_producer_statistics():
    """The wide GroupNorm as the step runs it: statistics from the producers' per-(row tile, channel) partials (built here with torch),
    beside the self-contained pair (statistics launch + apply launch).  Round 4, `profiles/r04h_groupnorm_merge_microbench.txt`, also had
    the round-3 gather merge and a 128-chunk geometry in this table (both removed from the library since)."""
    dev = "cuda:0"
    print(f"{'op':10s} {'B':>2s} {'side':>4s} {'C':>5s} {'tile':>4s} {'MB':>7s} {'own':>7s} {'pstats':>7s}  us")
    for B in (3, 4):
        # rows = side * side: ONE partial per channel — the merge costs nothing, what is left is the apply kernel's floor
        for side, C, rows in ((64, 320, 128), (64, 320, 256), (64, 320, 4096), (64, 640, 256), (64, 960, 256), (32, 640, 128), (32, 640, 64),
                              (32, 640, 1024), (32, 1280, 128), (32, 1280, 1024), (32, 1920, 128), (16, 1920, 64), (16, 2560, 64), (16, 2560, 256)):
            hw = side * side
            if not ops.groupnorm_uses_pstats(hw, C, 32):
                continue
            x = torch.randn(B, hw, C, device=dev)
            tiles = x.view(B, hw // rows, rows, C)
            st = torch.stack([tiles.sum(2), (tiles * tiles).sum(2)], dim=2).contiguous().view(-1)      # [B][tiles][2][C]
            g, b = torch.randn(C, device=dev).half(), torch.randn(C, device=dev).half()
            y = torch.zeros(B, side + 2, side + 2, C, dtype=torch.float16, device=dev)
            ws = torch.empty(ops.groupnorm_workspace_bytes(B, 32), dtype=torch.uint8, device=dev)
            res = []
            for pst in (None, [(st, rows, C)]):
                res.append(timed(lambda: ops.groupnorm(x, g, b, y, 32, 1e-5, True, ws, pstats=pst)))
            mb = B * hw * C * 6 / 1e6
            print(f"{'groupnorm':10s} {B:2d} {side:4d} {C:5d} {rows:4d} {mb:7.2f} " + " ".join(f"{u:7.2f}" for u in res))


def attention_d160():
    """Attention shapes of the step: head dim 160 (16x16 level: few workgroups, long key loops), 80 and 40."""
    dev = "cuda:0"
    print(f"{'op':10s} {'B':>2s} {'Nq':>5s} {'Nk':>5s} {'us':>8s} {'TFLOP/s':>8s}")
    for D, B, Nq, Nk in ((160, 3, 256, 768), (160, 4, 256, 256), (160, 3, 256, 256), (160, 3, 256, 77), (160, 3, 576, 2880),
                         (80, 3, 1024, 3072), (80, 4, 1024, 1024), (80, 3, 1024, 77), (40, 3, 4096, 12288), (40, 4, 4096, 4096),
                         (40, 4, 4096, 77)):
        C = 8 * D
        q = torch.randn(B, Nq, C, device=dev).half()
        k = torch.randn(B, Nk, C, device=dev).half()
        nk8 = (Nk + 7) // 8 * 8
        vt = torch.randn(B, C, nk8, device=dev).half()
        o = torch.empty_like(q)
        us = timed(lambda: ops.attention(q, k, vt, o, 8, D ** -0.5, nk=Nk), iters=10 if Nk > 4096 else 30)
        print(f"{f'attn_d{D}':10s} {B:2d} {Nq:5d} {Nk:5d} {us:8.2f} {4.0 * B * 8 * Nq * Nk * D / us / 1e6:8.1f}")


if __name__ == "__main__":
    if "--pstats" in sys.argv:
        groupnorm_with_producer_statistics()
        sys.exit(0)
    if "--attn" in sys.argv:
        attention_d160()
        sys.exit(0)
    main()
