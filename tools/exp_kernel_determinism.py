#!/usr/bin/env python
"""Stress a GEMM plan for run-to-run differences (development tool): every launch of the same problem must write the same bits.  Each
problem is launched REPS times, alternating with a co-running stream of other kernels so that the workgroup timing varies; outputs are
compared bit for bit against the first.

    python tools/exp_kernel_determinism.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402

ops.apply_env_options()
dev = torch.device("cuda:0")
F16, F32 = torch.float16, torch.float32


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    ws_b = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    # background load on a second stream: a big GEMM and a convolution, so that the tested launches share CUs with other kernels
    ba = torch.randn(8192, 1280, device=dev).half()
    bw = torch.randn(1280, 1280, device=dev).half()
    bo = torch.empty(8192, 1280, dtype=F16, device=dev)
    guard = torch.zeros(1, dtype=torch.int32, device=dev)
    # "polluters": a different launch right before each tested one (same stream, no synchronisation) so that the LDS / register
    # state a workgroup inherits varies from repeat to repeat — a kernel that reads LDS it never wrote shows up as a difference
    import random
    rnd = random.Random(0)
    pol = []
    for pm, pn, pk in [(4096, 320, 320), (2048, 640, 2560), (1024, 1280, 640), (512, 256, 1280)]:
        pa, pw_ = torch.randn(pm, pk, device=dev).half(), torch.randn(pn, pk, device=dev).half()
        po = torch.empty(pm, pn, dtype=F16, device=dev)
        for tl in (None, (64, 64, 4), (256, 128), (128, 128)):
            pol.append((pa, pw_, po, tl))

    def pollute():
        if "pollute" in sys.argv:
            pa, pw_, po, tl = pol[rnd.randrange(len(pol))]
            pa.mul_(1.0001)
            ops.gemm(pa, pw_, po, tile=tl, workspace=ws_b)

    cases = []
    for M, N, K in [(768, 1280, 1280), (3072, 640, 640), (192, 1280, 2560), (768, 1280, 2560), (3072, 640, 3200)]:
        for tile in [None, (64, 64, 4), (64, 128, 8), (128, 64)]:
            cases.append((M, N, K, tile))
    bad_total = 0
    for M, N, K, tile in cases:
        a = torch.randn(M, K, device=dev).half()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).half()
        res = torch.randn(M, N, device=dev)
        bias = torch.randn(N, device=dev).half()
        out = torch.empty(M, N, dtype=F32, device=dev)
        out2 = torch.empty(M, N, dtype=F16, device=dev)
        st = torch.empty(M, (N // 64 + 1) & ~1, 2, dtype=F32, device=dev)
        first = None
        bad = 0
        for r in range(reps):
            out.fill_(float("nan"))
            with torch.cuda.stream(side):
                if r % 2:
                    ops.gemm(ba, bw, bo, workspace=ws_b)
            pollute()
            ops.gemm(a, w, out, bias=bias, res1=res, out2=out2, ln_out=st, guard=guard, tile=tile, workspace=ws)
            torch.cuda.synchronize()
            cur = (out.clone(), out2.clone(), st[:, : N // 64].clone())
            if first is None:
                first = cur
            elif not all(torch.equal(x, y) for x, y in zip(cur, first)):
                bad += 1
        print(f"M{M} N{N} K{K} tile {tile}: {bad} of {reps} launches differ from the first", flush=True)
        bad_total += bad
    # paired launches (sg_gemm_pair_f16): q|k + V^T of one LayerNorm-folded operand, two plain projections of one operand (k3 + v3^T)
    from storygen_amd.repack import fold_layernorm
    for M, C in [(768, 1280), (3072, 640), (192, 1280), (48, 1280), (240, 640)]:
        for tile in [None, (64, 64, 4), (128, 64)]:
            x16 = torch.randn(M, C, device=dev).half()
            st = torch.zeros(M, (C // 64 + 1) & ~1, 2, dtype=F32, device=dev)
            st[:, :, 0] = torch.randn(M, (C // 64 + 1) & ~1, device=dev)
            st[:, :, 1] = 64.0 + torch.rand(M, (C // 64 + 1) & ~1, device=dev)
            g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
            wq = (torch.randn(2 * C, C, device=dev) / C ** 0.5).half()
            wv = (torch.randn(C, C, device=dev) / C ** 0.5).half()
            wf, cq, dq = fold_layernorm(wq, None, g, b)
            wvf, cv, dv = fold_layernorm(wv, None, g, b)
            y, yt = torch.empty(M, 2 * C, dtype=F16, device=dev), torch.empty(C, M, dtype=F16, device=dev)
            k3, v3 = torch.empty(M, C, dtype=F16, device=dev), torch.empty(C, M, dtype=F16, device=dev)
            ws2 = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
            for kind in ("ln pair", "plain pair"):
                first, bad = None, 0
                for r in range(reps):
                    with torch.cuda.stream(side):
                        if r % 2:
                            ops.gemm(ba, bw, bo, workspace=ws_b)
                    if kind == "ln pair":
                        y.fill_(float("nan")), yt.fill_(float("nan"))
                        pollute()
                        ops.gemm_pair(((x16, wf, y), dict(ln=(1, st, cq, dq, 1e-5), guard=guard, tile=tile)),
                                      ((wvf, x16, yt), dict(ln=(2, st, cv, dv, 1e-5), guard=guard)))
                        cur = (y.clone(), yt.clone())
                    else:
                        k3.fill_(float("nan")), v3.fill_(float("nan"))
                        pollute()
                        ops.gemm_pair(((x16, wv, k3), dict(workspace=ws, tile=tile)), ((wv, x16, v3), dict(workspace=ws2)))
                        cur = (k3.clone(), v3.clone())
                    torch.cuda.synchronize()
                    if first is None:
                        first = cur
                    elif not all(torch.equal(a_, b_) for a_, b_ in zip(cur, first)):
                        bad += 1
                print(f"{kind} M{M} C{C} tile {tile}: {bad} of {reps} launches differ from the first", flush=True)
                bad_total += bad
    # the text K / V^T projections (attn2: 77 tokens padded to 80 per sample, K = 768): ragged token counts on both sides of the pair
    for B, C in [(3, 1280), (3, 640), (3, 320), (20, 1280)]:
        for tile in [None, (64, 64, 4), (128, 64)]:
            x = torch.randn(B * 80, 768, device=dev).half()
            wk = (torch.randn(C, 768, device=dev) / 768 ** 0.5).half()
            wv = (torch.randn(C, 768, device=dev) / 768 ** 0.5).half()
            kt, vtt = torch.empty(B * 80, C, dtype=F16, device=dev), torch.empty(C, B * 80, dtype=F16, device=dev)
            ws2 = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
            first, bad = None, 0
            for r in range(reps):
                with torch.cuda.stream(side):
                    if r % 2:
                        ops.gemm(ba, bw, bo, workspace=ws_b)
                kt.fill_(float("nan")), vtt.fill_(float("nan"))
                pollute()
                ops.gemm_pair(((x, wk, kt), dict(workspace=ws, tile=tile)), ((wv, x, vtt), dict(workspace=ws2)))
                torch.cuda.synchronize()
                cur = (kt.clone(), vtt.clone())
                if first is None:
                    first = cur
                elif not all(torch.equal(a_, b_) for a_, b_ in zip(cur, first)):
                    bad += 1
            print(f"text K/V pair B{B} C{C} tile {tile}: {bad} of {reps} launches differ from the first; finite {bool(torch.isfinite(cur[0]).all() and torch.isfinite(cur[1]).all())}", flush=True)
            bad_total += bad
    print("TOTAL differing launches:", bad_total)


if __name__ == "__main__":
    main()
