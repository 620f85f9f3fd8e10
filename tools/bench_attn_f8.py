#!/usr/bin/env python
"""fp8 vs fp16 attention at the shapes of BASELINE config 5 (96x96 latent, R = 5: 46 080 context keys) and config 2 (development)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (B, Bk, Nq, Nk) in [(3, 2, 9216, 46080), (3, 3, 9216, 9216), (3, 2, 4096, 12288), (4, 4, 4096, 4096)]:
    H, D = 8, 40
    C = H * D
    q = torch.randn(B, Nq, C, device=dev).half()
    k = torch.randn(Bk, Nk, C, device=dev).half()
    vt = torch.randn(Bk, C, Nk, device=dev).half()
    o = torch.empty(B, Nq, C, dtype=torch.float16, device=dev)
    nb = ops.attention_f8_bytes(B, H, Nq, False) + ops.attention_f8_bytes(Bk, H, Nk, False) + ops.attention_f8_bytes(Bk, H, Nk, True) + 4096
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    fl = 4.0 * B * H * Nq * Nk * D
    t16 = timeit(lambda: ops.attention(q, k, vt, o, H, D ** -0.5))
    t8 = timeit(lambda: ops.attention_f8(q, k, vt, o, H, D ** -0.5, scratch))
    print(f"B{B} Bk{Bk} Nq{Nq} Nk{Nk}: fp16 {t16:8.1f} us ({fl / t16 / 1e6:6.0f} TFLOP/s)   fp8 incl. packing {t8:8.1f} us ({fl / t8 / 1e6:6.0f} TFLOP/s)",
          flush=True)
