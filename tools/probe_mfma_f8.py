#!/usr/bin/env python
"""Operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, unit scales) established ON THE DEVICE — the one piece of
information the fp8 attention path of BASELINE config 5 needs before its kernel can be written (the CDNA4 guide defers
the table to a file that is not in this image).  Run on an MI355X:

    python tools/probe_mfma_f8.py

1. hypothesis test: the natural map  A[i][k] <- lane i + 32 (k // 32), byte k % 32  (and B[k][j] likewise with j), checked
   with random fp8 operands against a float reference through the known C/D map
   (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = lane & 31);
2. if that fails: discovery by one-hot probes — the row of every A byte, the column of every B byte, and which A / B bytes
   share a k — printed as tables."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402

DEV = "cuda:0"
ONE = 0x38          # 1.0 in e4m3 (bias 7)


def d_matrix(raw):
    """[64 lanes][16 regs] -> D [32, 32] through the C/D map every 32x32 MFMA shares."""
    out = torch.empty(32, 32)
    raw = raw.cpu()
    for lane in range(64):
        for r in range(16):
            out[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31] = raw[lane, r]
    return out


def run(a, b):
    return d_matrix(ops.debug_mfma_f8(a.to(DEV).contiguous(), b.to(DEV).contiguous()))


def hypothesis_natural():
    g = torch.Generator().manual_seed(0)
    A = (torch.randn(32, 64, generator=g) * 0.5).to(torch.float8_e4m3fn)
    B = (torch.randn(64, 32, generator=g) * 0.5).to(torch.float8_e4m3fn)
    a, b = torch.zeros(64, 32, dtype=torch.uint8), torch.zeros(64, 32, dtype=torch.uint8)
    Ab, Bb = A.view(torch.uint8), B.view(torch.uint8)
    for k in range(64):
        for i in range(32):
            a[i + 32 * (k // 32), k % 32] = Ab[i, k]
            b[i + 32 * (k // 32), k % 32] = Bb[k, i]
    got, ref = run(a, b), A.float() @ B.float()
    err = float((got - ref).norm() / ref.norm())
    print(f"natural map (lane = row/col + 32 (k // 32), byte = k % 32): rel err {err:.2e}")
    return err < 1e-5


def discover():
    ones = torch.full((64, 32), ONE, dtype=torch.uint8)
    rows, cols = torch.zeros(64, 32, dtype=torch.long), torch.zeros(64, 32, dtype=torch.long)
    for lane in range(64):
        for byte in range(32):
            x = torch.zeros(64, 32, dtype=torch.uint8)
            x[lane, byte] = ONE
            rows[lane, byte] = int(run(x, ones).sum(1).argmax())       # A one-hot x B ones: its row lights up
            cols[lane, byte] = int(run(ones, x).sum(0).argmax())
    print("row of A[lane][byte] (lanes 0..3, 32..35):\n", rows[[0, 1, 2, 3, 32, 33, 34, 35]])
    print("col of B[lane][byte] (lanes 0..3, 32..35):\n", cols[[0, 1, 2, 3, 32, 33, 34, 35]])
    # k pairing for row 0 / column 0: which B byte of column 0 meets each A byte of row 0
    a_pos = [(l, y) for l in range(64) for y in range(32) if rows[l, y] == 0]
    b_pos = [(l, y) for l in range(64) for y in range(32) if cols[l, y] == 0]
    print(f"row 0 has {len(a_pos)} A bytes, column 0 has {len(b_pos)} B bytes")
    pairing = {}
    for (la, ya) in a_pos:
        a = torch.zeros(64, 32, dtype=torch.uint8)
        a[la, ya] = ONE
        lo, hi = 0, len(b_pos)
        while hi - lo > 1:                                               # binary search over the candidate B bytes
            mid = (lo + hi) // 2
            b = torch.zeros(64, 32, dtype=torch.uint8)
            for (lb, yb) in b_pos[lo:mid]:
                b[lb, yb] = ONE
            if float(run(a, b)[0, 0]) != 0.0:
                hi = mid
            else:
                lo = mid
        pairing[(la, ya)] = b_pos[lo]
    print("A (lane, byte) of row 0  <->  B (lane, byte) of column 0 sharing its k:")
    for k, v in sorted(pairing.items()):
        print("  ", k, "<->", v)


if __name__ == "__main__":
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    if hypothesis_natural():
        print("LAYOUT_OK natural")
    else:
        discover()
