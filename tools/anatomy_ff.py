#!/usr/bin/env python
"""Where the cycles of the fused GEGLU feed-forward kernel go (development tool, round 4): launches sg_ff_geglu_fused_f16 through its
instrumented instantiation (experiments library, sg_debug_ff_anatomy: s_memtime stamps around the phases of every hidden-chunk
iteration, summed per wave) for every variant (development option ff_variant: bit 0 = refill spread over the k-steps, bit 1 = W1 fragments
two k-steps ahead) and prints mean cycles per iteration and phase, next to the launch time of the plain kernel and of the two GEMM
launches it replaces.  MFMA issue alone is 60 x 32 = 1 920 cycles per iteration and wave.
Usage: python -m storygen_amd.build --experiments && python tools/anatomy_ff.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import _lib, build  # noqa: E402

if not os.path.exists(build.LIB_EXP):
    sys.exit("tools/anatomy_ff.py needs the experiments library: python -m storygen_amd.build --experiments")
_lib.LIB_PATH = build.LIB_EXP
from storygen_amd import ops  # noqa: E402
from storygen_amd.repack import ff_fused_pack, fold_layernorm, interleave_geglu  # noqa: E402

dev = torch.device("cuda:0")
NAMES = ["iters", "wait", "barrier", "issue+d1", "gemm1+geglu", "gemm2", "prologue", "total"]


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    C = 320
    g = torch.Generator().manual_seed(0)
    w1 = (torch.randn(8 * C, C, generator=g) * C ** -0.5).half().to(dev)
    b1 = torch.randn(8 * C, generator=g).half().to(dev)
    w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).half().to(dev)
    b2 = torch.randn(C, generator=g).half().to(dev)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).half().to(dev), (0.2 * torch.randn(C, generator=g)).half().to(dev)
    w1i, b1i = interleave_geglu(w1, b1)
    w1f, c1, d1 = fold_layernorm(w1i, b1i, gamma, beta)
    pack = ff_fused_pack(w1f.contiguous(), d1.contiguous(), w2)
    for M in (12288, 16384):
        x = torch.randn(M, C, device=dev) * 1.5
        out = torch.empty(M, C, dtype=torch.float16, device=dev)
        raw, lnst = x.half(), torch.zeros(M, (C // 64 + 1) & ~1, 2, device=dev)
        ffi = torch.empty(M, 4 * C, dtype=torch.float16, device=dev)
        us2 = timed(lambda: (ops.gemm(raw, w1f, ffi, epilogue=ops.EPI_GEGLU, ln=(1, lnst, c1, d1, 1e-5)), ops.gemm(ffi, w2, out, bias=b2, res1=x)))
        print(f"M{M}: the two GEMM launches {us2:6.1f} us")
        for var in range(4):
            ops.debug_set_option("ff_variant", var)
            us = timed(lambda: ops.ff_fused(x, pack, b2, out))
            prof = torch.zeros(8 * 4 * ((M + 127) // 128), dtype=torch.int64, device=dev)
            ops.ANATOMY = prof
            try:
                ops.ff_fused(x, pack, b2, out)
                torch.cuda.synchronize()
            finally:
                ops.ANATOMY = None
            rows = prof.view(-1, 8).double()
            it = rows[:, 0].sum()
            per = rows[:, 1:6].sum(0) / it
            print(f"  variant {var} (spread {var & 1}, deep {var >> 1 & 1}): {us:6.1f} us | per iteration: "
                  + " ".join(f"{n} {v:6.0f}" for n, v in zip(NAMES[1:6], per.tolist())) + f" = {per.sum():6.0f} cycles | prologue "
                  f"{rows[:, 6].mean():6.0f}  total {rows[:, 7].mean():8.0f} (max {rows[:, 7].max():8.0f})", flush=True)
        ops.debug_set_option("ff_variant", 0)


if __name__ == "__main__":
    main()
