#!/usr/bin/env python
"""Backward attention of the stage-2 training step (BASELINE config 4) per shape: sg_attn_bwd_dq_f16 / sg_attn_bwd_dkv_f16 timed with
events, back to back (development tool; the shapes are the ones `bench.py --train-step` runs at batch 4 with 3 reference frames).
Usage: python tools/bench_attn_bwd.py [path/to/another/libstorygen_hip.so] [reps=20]
TFLOP/s are EXECUTED flops (dQ pass: S, dP, dQ = 6 B H Nq Nk D; dK/dV pass: S, dP, dK, dV = 8 B H Nq Nk D)."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from storygen_amd import _lib  # noqa: E402

args = [a for a in sys.argv[1:]]
reps = 20
for a in list(args):
    if a.startswith("reps="):
        reps = int(a[5:]); args.remove(a)
if args:
    _lib.LIB_PATH = os.path.abspath(args[0])
import torch  # noqa: E402
from storygen_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [  # B, H, D, Nq, Nk, dkv too?
    (4, 8, 40, 4096, 4096, True), (4, 8, 40, 4096, 12288, True), (4, 8, 40, 4096, 77, False),
    (4, 8, 80, 1024, 1024, True), (4, 8, 80, 1024, 3072, True), (4, 8, 80, 1024, 77, False),
    (4, 8, 160, 256, 256, True), (4, 8, 160, 256, 768, True), (4, 8, 160, 64, 192, True),
]


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


print(f"library: {_lib.LIB_PATH}")
print(f"{'shape':44s} {'dq us':>9s} {'TFLOP/s':>8s} {'dkv us':>9s} {'TFLOP/s':>8s}")
tot = 0.0
for B, H, D, Nq, Nk, kv in SHAPES:
    C = H * D
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).half().to(dev)  # noqa: E731
    q, k, v, do = r(B, Nq, C), r(B, Nk, C), r(B, Nk, C), r(B, Nq, C)
    o, lse2 = torch.empty_like(q), torch.empty(B, H, Nq, dtype=torch.float32, device=dev)
    scale = D ** -0.5
    Nk8 = (Nk + 7) // 8 * 8                               # transposed operands: rows padded to 8 keys with finite data

    def tr(t, n8):
        out = torch.zeros(B, C, n8, dtype=torch.float16, device=dev)
        out[:, :, :t.shape[1]] = t.transpose(1, 2)
        return out[:, :, :t.shape[1]]
    ops.attention_lse(q, k, tr(v, Nk8), o, lse2, H, scale)
    ld2 = torch.empty(B, H, Nq, 2, dtype=torch.float32, device=dev)
    ops.attention_bwd_prep(o, do, lse2, ld2, H)
    kt, qt, dot = tr(k, Nk8), tr(q, Nq), tr(do, Nq)
    dq = torch.empty_like(q)
    dkt, dvt = (torch.empty(B, C, Nk, dtype=torch.float16, device=dev) for _ in range(2))
    t_dq = timed(lambda: ops.attention_bwd_dq(q, k, kt, v, do, ld2, dq, H, scale))
    line = f"B{B} H{H} D{D} Nq{Nq} Nk{Nk}".ljust(44) + f" {t_dq:9.1f} {6.0 * B * H * Nq * Nk * D / t_dq * 1e-6:8.1f}"
    tot += t_dq
    if kv:
        t_kv = timed(lambda: ops.attention_bwd_dkv(q, qt, k, v, do, dot, ld2, dkt, dvt, H, scale))
        line += f" {t_kv:9.1f} {8.0 * B * H * Nq * Nk * D / t_kv * 1e-6:8.1f}"
        tot += t_kv
    print(line, flush=True)
print(f"sum of the listed launches: {tot / 1e3:.3f} ms")
