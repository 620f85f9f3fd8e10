#!/usr/bin/env python
"""Does any GEMM plan write outside its output (development tool)?  Outputs are views into sentinel-filled buffers with guard rows above
and below and guard columns to the right; after the launch everything outside the view must still hold the sentinel.  Shapes: the
swapped-operand / ragged problems of the paired launches (V^T = Wv X^T with few tokens), the producers with second outputs and
LayerNorm partials, split-K."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402
from storygen_amd.repack import fold_layernorm  # noqa: E402

ops.apply_env_options()
dev = torch.device("cuda:0")
F16, F32 = torch.float16, torch.float32
SENT = 12345.0


def guarded(rows, cols, dtype, extra_cols=16, guard_rows=64):
    big = torch.full((rows + 2 * guard_rows, cols + extra_cols), SENT, dtype=dtype, device=dev)
    return big, big[guard_rows:guard_rows + rows, :cols]


def intact(big, rows, cols, guard_rows=64):
    ok = bool((big[:guard_rows] == SENT).all() and (big[guard_rows + rows:] == SENT).all() and (big[:, cols:] == SENT).all())
    return ok


def main():
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    guard = torch.zeros(1, dtype=torch.int32, device=dev)
    bad = 0
    for tile in [(64, 64, 4), (64, 128, 8), None, (128, 64), (128, 128)]:
        for M, C in [(192, 1280), (48, 1280), (768, 640), (240, 768), (200, 320), (72, 1280)]:
            x16 = torch.randn(M, C, device=dev).half()
            st = torch.zeros(M, (C // 64 + 1) & ~1, 2, dtype=F32, device=dev)
            st[:, :, 1] = 64.0
            g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
            wv = (torch.randn(1280, C, device=dev) / C ** 0.5).half()
            wvf, cv, dv = fold_layernorm(wv, None, g, b)
            for kind in ("ln2", "swapped", "ln1", "plain", "producer", "split"):
                try:
                    if kind in ("ln2", "swapped"):
                        big, o = guarded(1280, M, F16, extra_cols=8)
                        if kind == "ln2":
                            ops.gemm(wvf, x16, o, ln=(2, st, cv, dv, 1e-5), guard=guard, tile=tile)
                        else:
                            ops.gemm(wv, x16, o, tile=tile, split_k=1)
                        ok = intact(big, 1280, M)
                    elif kind in ("ln1", "plain"):
                        big, o = guarded(M, 1280, F16)
                        if kind == "ln1":
                            ops.gemm(x16, wvf, o, ln=(1, st, cv, dv, 1e-5), guard=guard, tile=tile)
                        else:
                            ops.gemm(x16, wv, o, tile=tile, split_k=1)
                        ok = intact(big, M, 1280)
                    elif kind == "producer":
                        big, o = guarded(M, 1280, F32)
                        big2, o2 = guarded(M, 1280, F16)
                        bigs = torch.full((M + 128, 20, 2), SENT, dtype=F32, device=dev)
                        sto = bigs[64:64 + M]
                        res = torch.randn(M, 1280, device=dev)
                        ops.gemm(x16, wv, o, res1=res, out2=o2, ln_out=sto, guard=guard, tile=tile, workspace=ws)
                        ok = intact(big, M, 1280) and intact(big2, M, 1280) and bool((bigs[:64] == SENT).all() and (bigs[64 + M:] == SENT).all())
                    else:
                        big, o = guarded(M, 1280, F16)
                        ops.gemm(x16, wv, o, tile=tile, split_k=2, workspace=ws)
                        ok = intact(big, M, 1280)
                    torch.cuda.synchronize()
                except Exception as e:
                    print(f"tile {tile} M{M} C{C} {kind}: n/a ({str(e)[:80]})")
                    continue
                if not ok:
                    bad += 1
                    print(f"tile {tile} M{M} C{C} {kind}: WRITES OUTSIDE ITS OUTPUT", flush=True)
    print("launches that wrote outside their output:", bad)


if __name__ == "__main__":
    main()
