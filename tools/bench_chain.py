#!/usr/bin/env python
"""In-graph cost of the main pass's small GEMMs (development tool).

A UNet pass is a dependent chain of a few hundred launches inside one hipGraph: what a launch costs THERE (kernel boundary + ramp +
first-load latency + its slabs + store drain) is what the step pays, not what back-to-back independent launches of tools/bench_gemm.py
read.  This tool captures a linear chain of N copies of one launch (same stream = dependent graph nodes), replays it, and prints the
time per node — for each GEMM shape of the batch-3 main pass with the epilogue it carries there, under each tile plan given:

    python tools/bench_chain.py [default] [lat] [128x64] ...

`default` = tile table + cost model; `lat` = the 32x32-per-wave deep-ring kernel (tile 64x64, 4 waves); `BMxBN` = a forced tile.
The first line is the floor: a chain of empty-ish launches (a 1-row copy).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402

ops.apply_env_options()
dev = torch.device("cuda:0")
F16, F32 = torch.float16, torch.float32
NODES = 40


def chain_us(fn, nodes=NODES, reps=8):
    """µs per node of a captured linear chain of `nodes` calls of fn."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(nodes):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            g.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * nodes) * 1e3


def chain_us_rot(fns, reps=8):
    """As chain_us, but the chain is the given list of DIFFERENT calls (e.g. one launch per weight copy: cold operands)."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fns[:2]:
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for f in fns:
                f()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            g.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * len(fns)) * 1e3


def cold_weights():
    """What a weight panel that is NOT in L2 / MALL costs a dependent launch: the same GEMM chained over 1 (warm), 16 and 160 distinct
    weight copies (160 x 3.3 MB = 524 MB > the 256 MB Infinity Cache: every launch streams its weights from HBM, as in the step,
    where 1.8 GB of weights pass between two uses of a layer)."""
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    print("cold weights: us per graph node by number of distinct weight copies in the chain")
    for M, N, K in [(768, 1280, 1280), (768, 1280, 2560), (3072, 640, 640), (192, 1280, 1280)]:
        a = torch.randn(M, K, device=dev).half()
        out = torch.empty(M, N, dtype=F32, device=dev)
        res = torch.randn(M, N, device=dev)
        row = f"M{M} N{N} K{K} res".ljust(34)
        for copies in (1, 16, 160):
            wsx = [(torch.randn(N, K, device=dev) / K ** 0.5).half() for _ in range(copies)]
            fns = [(lambda w=w: ops.gemm(a, w, out, res1=res, workspace=ws)) for w in wsx] * (160 // copies)
            row += f"{chain_us_rot(fns, reps=4):12.2f}"
            del wsx
        print(row, flush=True)


def split_of(name):
    """'lat/s2' -> split_k 2 (0 = the plan's own choice)."""
    return int(name.split("/s")[1]) if "/s" in name else 0


def plan_of(name):
    name = name.split("/s")[0]
    if name == "default":
        return None
    if name == "lat":
        return (64, 64, 4)
    if name == "latw":
        return (64, 128, 8)
    bm, bn = name.split("x")
    return (int(bm), int(bn))


def main():
    if "--cold" in sys.argv:
        return cold_weights()
    plans = [a for a in sys.argv[1:] if not a.startswith("-")] or ["default"]
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    ws2 = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    guard = torch.zeros(1, dtype=torch.int32, device=dev)
    one = torch.zeros(1, 1, 64, dtype=F16, device=dev)
    one2 = torch.zeros(1, 1, 64, dtype=F16, device=dev)
    print(f"floor (copy of one 64-element row): {chain_us(lambda: ops.copy_rows(one2, one)):.2f} us per node")
    # (M, N, K, kind): kind = plain fp16 out | 'res' fp32 out + fp32 residual + fp16 copy + LayerNorm partials (proj_in / to_out1) |
    #                  'res2' the (a2 + h) + (a3 + h) combine | 'ln' LayerNorm-folded consumer | 'stats' proj_out (+ GroupNorm partials)
    shapes = [(768, 1280, 1280, "res"), (768, 1280, 1280, "ln"), (768, 1280, 2560, "res2"), (768, 1280, 5120, "stats"), (768, 2560, 1280, "plain"),
              (3072, 640, 640, "res"), (3072, 640, 640, "ln"), (3072, 640, 1280, "res2"), (3072, 640, 2560, "stats"), (3072, 1280, 640, "plain"),
              (12288, 320, 320, "res"), (12288, 320, 320, "ln"), (12288, 320, 640, "res2"), (12288, 640, 320, "plain"),
              (192, 1280, 1280, "res"), (192, 1280, 2560, "res2"), (192, 1280, 5120, "stats")]
    print(f"{'shape':34s}" + "".join(f"{p:>12s}" for p in plans) + "   (us per graph node)")
    for M, N, K, kind in shapes:
        a = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        bias = torch.randn(N, device=dev).half()
        res = torch.randn(M, N, device=dev)
        kw = {}
        if kind == "plain":
            out = torch.empty(M, N, dtype=F16, device=dev)
        elif kind in ("res", "res2", "stats"):
            out = torch.empty(M, N, dtype=F32, device=dev)
            kw = dict(bias=bias, res1=res)
            if kind == "res":
                kw.update(out2=torch.empty(M, N, dtype=F16, device=dev), ln_out=torch.empty(M, (N // 64 + 1) & ~1, 2, dtype=F32, device=dev), guard=guard)
            if kind == "res2":
                kw.update(res2=res)
        else:
            out = torch.empty(M, N, dtype=F16, device=dev)
            lst = torch.zeros(M, (K // 64 + 1) & ~1, 2, dtype=F32, device=dev)
            lst[:, :, 1] = 64.0
            kw = dict(ln=(1, lst, torch.zeros(N, device=dev), torch.zeros(N, device=dev), 1e-5), guard=guard)
        row = f"{f'M{M} N{N} K{K} {kind}':34s}"
        for name in plans:
            tile = plan_of(name)
            kk = dict(kw)
            if kind == "stats":
                hw = 64 if M == 192 else M // 3
                buf = torch.empty(2 * (M // 64) * N, dtype=F32, device=dev)
                try:
                    rows = ops.gemm_stats_rows(a, w, out, stats=(buf, hw), tile=tile, workspace=ws, **kk)
                except Exception:
                    rows = 0
                if rows:
                    kk["stats"] = (buf, hw)
            try:
                if kind == "ln" and split_of(name) > 1:
                    raise ValueError("a folded LayerNorm does not split K")
                us = chain_us(lambda: ops.gemm(a, w, out, tile=tile, workspace=ws, split_k=split_of(name), **kk))
                row += f"{us:12.2f}"
            except Exception as e:     # a plan this launch cannot take
                row += f"{'n/a':>12s}"
                if os.environ.get("SG_VERBOSE"):
                    print(e)
        print(row, flush=True)
    # the 3x3 convolutions of the 8x8 / 16x16 levels (bias + fp32 residual; split-K second pass included where the plan splits)
    print(f"{'conv3x3':34s}" + "".join(f"{p:>12s}" for p in plans))
    for B, H, W, Ci, Co, stride in [(3, 8, 8, 1280, 1280, 1), (3, 8, 8, 2560, 1280, 1), (3, 16, 16, 1280, 1280, 1), (3, 16, 16, 2560, 1280, 1),
                                    (3, 16, 16, 1280, 1280, 2), (3, 32, 32, 640, 640, 1), (3, 32, 32, 640, 640, 2)]:
        xp = torch.zeros(B, H + 2, W + 2, Ci, dtype=F16, device=dev)
        xp[:, 1:-1, 1:-1] = torch.randn(B, H, W, Ci, device=dev).half()
        w = (torch.randn(Co, 3, 3, Ci, device=dev) / (9 * Ci) ** 0.5).half()
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        out = torch.empty(B, Ho, Wo, Co, dtype=F32, device=dev)
        res = torch.randn(B, Ho, Wo, Co, device=dev)
        bias = torch.randn(Co, device=dev).half()
        row = f"{f'B{B} {H}x{W} {Ci}->{Co} s{stride}':34s}"
        for name in plans:
            try:
                us = chain_us(lambda: ops.conv3x3(xp, w, out, stride=stride, bias=bias, res1=res, workspace=ws, x_padded=True, tile=plan_of(name)))
                row += f"{us:12.2f}"
            except Exception as e:
                row += f"{'n/a':>12s}"
                if os.environ.get("SG_VERBOSE"):
                    print(e)
        print(row, flush=True)


if __name__ == "__main__":
    main()
