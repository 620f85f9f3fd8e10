#!/usr/bin/env python
"""Run-to-run differences of ONE launch under real concurrency (development tool): the tested GEMM is launched REPS times back to back
into REPS different output buffers while a second stream keeps the chip busy with other kernels (no synchronisation in between), then
all outputs are compared bit for bit.  Covers the problem kinds that only occur as the second problem of a paired launch: the
columns-are-tokens LayerNorm fold (V^T = Wv LN(x)^T) and plain swapped-operand projections.

    python tools/exp_concurrent_determinism.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402
from storygen_amd.repack import fold_layernorm  # noqa: E402

ops.apply_env_options()
dev = torch.device("cuda:0")
F16, F32 = torch.float16, torch.float32


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    side = torch.cuda.Stream()
    ws_b = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    ba = torch.randn(12288, 640, device=dev).half()
    bw = torch.randn(640, 640, device=dev).half()
    bo = torch.empty(12288, 640, dtype=F16, device=dev)
    xp = torch.zeros(3, 34, 34, 640, dtype=F16, device=dev)
    cw = torch.randn(640, 3, 3, 640, device=dev).half()
    co = torch.empty(3, 32, 32, 640, dtype=F16, device=dev)
    guard = torch.zeros(1, dtype=torch.int32, device=dev)
    total_bad = 0
    for tile in [(64, 64, 4), None, (128, 64)]:
        for M, C in [(768, 640), (192, 1280), (48, 1280), (240, 768)]:
            x16 = torch.randn(M, C, device=dev).half()
            st = torch.zeros(M, (C // 64 + 1) & ~1, 2, dtype=F32, device=dev)
            st[:, :, 0] = torch.randn(M, (C // 64 + 1) & ~1, device=dev)
            st[:, :, 1] = 64.0 + torch.rand(M, (C // 64 + 1) & ~1, device=dev)
            g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
            wv = (torch.randn(C if C in (640, 1280) else 1280, C, device=dev) / C ** 0.5).half()      # V^T = Wv x^T: C rows
            wvf, cv, dv = fold_layernorm(wv, None, g, b)
            for kind in ("ln2 (columns are tokens)", "swapped plain", "ln1 (rows are tokens)"):
                Cr = wv.shape[0]
                outs = [torch.full((Cr, M) if kind != "ln1 (rows are tokens)" else (M, Cr), float("nan"), dtype=F16, device=dev) for _ in range(reps)]
                torch.cuda.synchronize()
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):          # background: ~reps * 3 launches of a GEMM and a convolution
                    for _ in range(reps):
                        ops.gemm(ba, bw, bo, workspace=ws_b)
                        ops.conv3x3(xp, cw, co, workspace=ws_b, x_padded=True)
                for o in outs:
                    if kind.startswith("ln2"):
                        ops.gemm(wvf, x16, o, ln=(2, st, cv, dv, 1e-5), guard=guard, tile=tile)
                    elif kind.startswith("swapped"):
                        ops.gemm(wv, x16, o, tile=tile, split_k=1)
                    else:
                        ops.gemm(x16, wvf, o, ln=(1, st, cv, dv, 1e-5), guard=guard, tile=tile)
                torch.cuda.synchronize()
                bad = sum(0 if torch.equal(o, outs[0]) else 1 for o in outs[1:])
                fin = all(bool(torch.isfinite(o).all()) for o in outs)
                print(f"tile {tile} M{M} C{C} {kind}: {bad} of {reps - 1} differ from the first{'' if fin else '  NON-FINITE'}", flush=True)
                total_bad += bad
    print("TOTAL differing launches:", total_bad)


if __name__ == "__main__":
    main()
