#!/usr/bin/env python
"""Runs a few representative launches of the hot kernels in isolation so that rocprofv3 --pmc can attribute counters
per dispatch (development tool; see profiles/*pmc*).  Usage: rocprofv3 --pmc <counters> --kernel-trace
--output-format csv -d out -- python tools/pmc_kernels.py [attn|conv|gemm|gn|ff ...]
Round 5: also the batch-20 shapes of the batched reference pass (ref_ahead 5) and the fused feed-forward kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402

ops.apply_env_options()      # SG_* development variables -> sg_debug_set_option

dev = torch.device("cuda:0")
which = set(sys.argv[1:]) or {"attn", "conv", "gemm", "gn", "ff"}
REP = 3


def rnd(*shape, scale=1.0, dtype=torch.float16):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
if "attn" in which:
    for (B, Bk, Nq, Nk, D) in [(3, 2, 4096, 12288, 40), (4, 4, 4096, 4096, 40), (3, 2, 1024, 3072, 80), (20, 20, 4096, 4096, 40),
                               (3, 2, 256, 768, 160)]:
        C = 8 * D
        q, k, vt = rnd(B, Nq, C), rnd(Bk, Nk, C), rnd(Bk, C, Nk)
        o = torch.empty(B, Nq, C, dtype=torch.float16, device=dev)
        for _ in range(REP):
            ops.attention(q, k, vt, o, 8, D ** -0.5)
if "conv" in which:
    for (B, H, Ci, Co) in [(4, 64, 320, 320), (4, 32, 640, 640), (4, 16, 1280, 1280), (4, 64, 640, 320), (20, 64, 320, 320),
                           (20, 32, 640, 640), (20, 16, 1280, 1280), (3, 8, 1280, 1280)]:
        xp = torch.zeros(B, H + 2, H + 2, Ci, dtype=torch.float16, device=dev)
        xp[:, 1:-1, 1:-1] = rnd(B, H, H, Ci)
        w = rnd(Co, 3, 3, Ci, scale=(9 * Ci) ** -0.5)
        out = torch.empty(B, H, H, Co, dtype=torch.float32, device=dev)
        res = rnd(B, H, H, Co, dtype=torch.float32)
        for _ in range(REP):
            ops.conv3x3(xp, w, out, bias=rnd(Co), res1=res, workspace=ws, x_padded=True)
if "gemm" in which:
    for (M, N, K, geglu) in [(16384, 320, 320, False), (4096, 640, 640, False), (1024, 1280, 1280, False), (16384, 2560, 320, True),
                             (1024, 10240, 1280, True), (4096, 640, 2560, False), (81920, 320, 320, False), (20480, 5120, 640, True),
                             (768, 1280, 1280, False)]:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        if geglu:
            out = torch.empty(M, N // 2, dtype=torch.float16, device=dev)
            for _ in range(REP):
                ops.gemm(a, w, out, bias=rnd(N), epilogue=ops.EPI_GEGLU, workspace=ws)
        else:
            out = torch.empty(M, N, dtype=torch.float32, device=dev)
            res = rnd(M, N, dtype=torch.float32)
            for _ in range(REP):
                ops.gemm(a, w, out, bias=rnd(N), res1=res, workspace=ws)
if "gn" in which:
    for (B, HW, C) in [(4, 4096, 320), (4, 1024, 640), (4, 256, 1280)]:
        x = rnd(B, HW, C, dtype=torch.float32)
        y = torch.empty(B, HW, C, dtype=torch.float16, device=dev)
        wsg = torch.empty(ops.groupnorm_workspace_bytes(B, 32), dtype=torch.uint8, device=dev)
        for _ in range(REP):
            ops.groupnorm(x, rnd(C), rnd(C), y, 32, 1e-5, True, wsg)
if "ff" in which:
    from storygen_amd.repack import ff_fused_pack, fold_layernorm, interleave_geglu
    C = 320
    w1, b1 = interleave_geglu(rnd(8 * C, C, scale=C ** -0.5), rnd(8 * C))
    w1f, _, d1 = fold_layernorm(w1, b1, rnd(C) + 1.0, rnd(C))
    pack = ff_fused_pack(w1f.contiguous(), d1.contiguous(), rnd(C, 4 * C, scale=(4 * C) ** -0.5))
    for M in (12288, 81920):
        x = rnd(M, C, dtype=torch.float32)
        y = torch.empty(M, C, dtype=torch.float16, device=dev)
        for _ in range(REP):
            ops.ff_fused(x, pack, rnd(C), y)
torch.cuda.synchronize()
print("done")
