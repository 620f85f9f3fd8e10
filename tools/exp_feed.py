#!/usr/bin/env python
"""Decomposition / ring-depth experiment for sg_gemm_f16 / sg_conv3x3 (development tool, round 2).

For each layer shape: every tile x split-K combination, COLD operands (a rotation of operand sets larger than the 256 MB
Infinity Cache, like the real step where each layer's weights were last touched one step ago) and warm (one set).
Run once per ring depth:  SG_STAGES=3|4|2 python tools/exp_feed.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402

ops.apply_env_options()      # SG_* development variables -> sg_debug_set_option

dev = torch.device("cuda:0")
TILES = [(256, 128), (128, 128), (256, 64), (128, 64), (64, 128), (64, 64)]
SPLITS = [1, 2, 3, 4, 6, 8]


def timeit(fns, n=24):
    for f in fns[:3]:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(30_000_000)
    a.record()
    for i in range(n):
        fns[i % len(fns)]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def sweep(name, flops, make, nsets, KT):
    sets = [make() for _ in range(nsets)]
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    res = {}
    for t in TILES:
        for s in SPLITS:
            if s > 1 and KT // s < 2:
                continue
            try:
                cold = timeit([(lambda st=st: st(t, s, ws)) for st in sets])
                warm = timeit([lambda: sets[0](t, s, ws)])
            except Exception as e:  # noqa: BLE001
                res[(t, s)] = None
                continue
            res[(t, s)] = (cold, warm)
    ok = {k: v for k, v in res.items() if v}
    best = min(ok, key=lambda k: ok[k][0])
    auto_c = timeit([(lambda st=st: st(None, 0, ws)) for st in sets])
    auto_w = timeit([lambda: sets[0](None, 0, ws)])
    print(f"{name:34s} auto cold {auto_c:6.1f} us ({flops / auto_c / 1e6:5.0f} TF) warm {auto_w:6.1f} | best cold {best[0][0]}x{best[0][1]}/s{best[1]} "
          f"{ok[best][0]:6.1f} us ({flops / ok[best][0] / 1e6:5.0f} TF) warm {ok[best][1]:6.1f}", flush=True)
    for t in TILES:
        row = f"    {t[0]:3d}x{t[1]:<3d} cold|warm:"
        for s in SPLITS:
            v = res.get((t, s))
            row += f"  s{s}: " + (f"{v[0]:5.1f}|{v[1]:5.1f}" if v else "    -      ")
        print(row, flush=True)


def main():
    only = sys.argv[1:] or None
    gemms = [("gemm M12288 N320 K320 f32+res", 12288, 320, 320, True, 0), ("gemm M3072 N640 K640 f32+res", 3072, 640, 640, True, 0),
             ("gemm M768 N1280 K1280 f32+res", 768, 1280, 1280, True, 0), ("gemm M768 N1280 K5120 f16", 768, 1280, 5120, False, 0),
             ("gemm M3072 N5120 K640 geglu", 3072, 5120, 640, False, 1), ("gemm M12288 N640 K320 f16", 12288, 640, 320, False, 0),
             ("gemm M192 N1280 K1280 f32+res", 192, 1280, 1280, True, 0)]
    for name, M, N, K, f32, epi in gemms:
        if only and not any(o in name for o in only):
            continue
        per = 2 * (M * K + N * K) + M * N * (8 if f32 else 2)
        nsets = max(2, min(24, (300 << 20) // per + 1))

        def make():
            a = torch.randn(M, K, device=dev).half()
            w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
            n_out = N // 2 if epi else N
            out = torch.empty(M, n_out, dtype=torch.float32 if f32 else torch.float16, device=dev)
            r = torch.randn(M, n_out, device=dev) if f32 else None
            b = torch.randn(N, device=dev).half()

            def run(t, s, ws):
                ops.gemm(a, w, out, bias=b, res1=r, epilogue=epi, split_k=s, workspace=ws, tile=t)
            return run
        sweep(name, 2.0 * M * N * K, make, nsets, K // 64)
    convs = [("conv B3 64x64 320->320", 3, 64, 64, 320, 320), ("conv B3 32x32 640->640", 3, 32, 32, 640, 640),
             ("conv B3 16x16 1280->1280", 3, 16, 16, 1280, 1280), ("conv B3 8x8 1280->1280", 3, 8, 8, 1280, 1280),
             ("conv B4 64x64 640->320", 4, 64, 64, 640, 320)]
    for name, B, H, W, Ci, Co in convs:
        if only and not any(o in name for o in only):
            continue
        per = 2 * (B * (H + 2) * (W + 2) * Ci + Co * 9 * Ci) + B * H * W * Co * 8
        nsets = max(2, min(24, (300 << 20) // per + 1))

        def make():
            xp = torch.zeros(B, H + 2, W + 2, Ci, dtype=torch.float16, device=dev)
            xp[:, 1:-1, 1:-1] = torch.randn(B, H, W, Ci, device=dev).half()
            w = (torch.randn(Co, 3, 3, Ci, device=dev) / (9 * Ci) ** 0.5).half()
            out = torch.empty(B, H, W, Co, dtype=torch.float32, device=dev)
            r = torch.randn(B, H, W, Co, device=dev)
            b = torch.randn(Co, device=dev).half()

            def run(t, s, ws):
                ops.conv3x3(xp, w, out, bias=b, res1=r, split_k=s, workspace=ws, x_padded=True, tile=t)
            return run
        sweep(name, 2.0 * B * H * W * Co * 9 * Ci, make, nsets, 9 * Ci // 64)


if __name__ == "__main__":
    main()
