#!/usr/bin/env python
"""Per-shape lower bounds for the MFMA-class launches of one denoising step, next to their measured in-situ times
(input: the table tools/profile_step.py prints, e.g. profiles/r01j_per_shape.txt).  For every GEMM / conv / attention shape:

  t_mfma  = FLOPs / 2.5 PFLOP/s                                  (dense fp16 MFMA peak, 256 CUs)
  t_hbm   = minimal operand + result bytes / 8 TB/s
  t_feed  = bytes the tiles must pull into LDS / (active CUs x 18.5 B/clk x 2.4 GHz)   (measured L2->LDS feed rate of the
            mainloop, HISTORY §5.2; active CUs = min(256, tiles x split) with the 256x128 tile unless the shape is smaller)
  t_lds   = fragment reads (1 KiB per 64x64x16 MFMA step per wave = FLOPs / 32 bytes) + the DMA writes, at 128 B/clk/CU on the
            active CUs
  floor   = 5 us (launch + pipeline fill + epilogue of a single-wave-of-tiles kernel)

and which of them binds.  measured / max(bounds) near 1 means the kernel sits on a structural bound of its current design
(the lever is then the design: tile shape, operand sharing, batching); >> 1 means latency / scheduling slack inside the
kernel.  A development aid for choosing what to work on; the model is deliberately crude.

    python tools/shape_bounds.py profiles/r01j_per_shape.txt
"""
import math
import re
import sys

CUS, CLK = 256, 2.4e9
PEAK, HBM, FEED, LDS = 2.5e15, 8e12, 18.5, 128.0


def tiles_of(M, N, bm=256, bn=128):
    bm = bm if M >= bm else max(64, 1 << (M - 1).bit_length())
    bn = bn if N >= bn else 64
    return math.ceil(M / bm) * math.ceil(N / bn), bm, bn


def gemm_bounds(M, N, K, out_bytes=2):
    flops = 2.0 * M * N * K
    nt, bm, bn = tiles_of(M, N)
    split = max(1, min(16, CUS // max(nt, 1))) if nt < CUS // 2 and K >= 1280 else 1
    active = min(CUS, nt * split)
    rounds = math.ceil(nt * split / CUS)
    per_tile = (bm + bn) * (K / split) * 2.0
    t_feed = rounds * per_tile / (FEED * CLK)
    lds_bytes = flops / 32.0 + (bm + bn) * K * 2.0 * nt
    t_lds = lds_bytes / (active * LDS * CLK)
    t_hbm = (2.0 * (M * K + N * K) + out_bytes * M * N) / HBM
    return flops, dict(mfma=flops / PEAK, hbm=t_hbm, feed=t_feed, lds=t_lds, floor=5e-6), f"{bm}x{bn} x{split}"


def parse(path):
    rows = []
    for ln in open(path):
        m = re.match(r"^(gemm|conv3x3|attention_d\d+)\s+(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m:
            rows.append((m.group(1), m.group(2).strip(), int(m.group(3)), float(m.group(4)), float(m.group(6))))
    return rows


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r01j_per_shape.txt"
    rows = parse(path)
    print(f"{'family':13s} {'shape':32s} {'n':>3s} {'us':>7s} | {'mfma':>6s} {'hbm':>6s} {'feed':>6s} {'lds':>6s} | {'bound':>5s} {'meas/bound':>10s}  tile")
    tot_meas = tot_bound = 0.0
    agg = {}
    for fam, shape, n, ms, us in rows:
        if fam == "gemm":
            M, N, K = (int(x) for x in re.match(r"M(\d+) N(\d+) K(\d+)", shape).groups())
            _, b, tile = gemm_bounds(M, N, K, 4)
        elif fam == "conv3x3":
            B, H, W, ci, co = (int(x) for x in re.match(r"B(\d+) (\d+)x(\d+) (\d+)->(\d+)", shape).groups())
            _, b, tile = gemm_bounds(B * H * W, co, 9 * ci, 4)       # the label carries the OUTPUT resolution
        else:
            D = int(fam.split("_d")[1])
            B, Hh, nq, nk = (int(x) for x in re.match(r"B(\d+) H(\d+) Nq(\d+) Nk(\d+)", shape).groups())
            flops = 4.0 * B * Hh * nq * nk * D
            pad = (math.ceil(D / 16) * 16 + math.ceil(D / 32) * 32) / (2.0 * D)     # MFMA work incl. head-dim padding
            wgs = B * Hh * math.ceil(nq / 128)
            active = min(CUS, wgs)
            kv_bytes = wgs * nk * D * 2.0 * 2                                         # every workgroup streams K and V^T once
            b = dict(mfma=flops * pad / PEAK, hbm=(2.0 * B * Hh * D * (2 * nq + 2 * nk)) / HBM,
                     feed=math.ceil(wgs / CUS) * (nk * D * 4.0) / (FEED * CLK) if wgs <= CUS else kv_bytes / (CUS * FEED * CLK),
                     lds=(flops * pad / 32.0 + kv_bytes) / (active * LDS * CLK), floor=5e-6)
            tile = f"{wgs} wg"
        which = max(b, key=b.get)
        bound = b[which]
        tot_meas += n * us
        tot_bound += n * bound * 1e6
        a = agg.setdefault(which, [0.0, 0.0])
        a[0] += n * us
        a[1] += n * bound * 1e6
        print(f"{fam:13s} {shape:32s} {n:3d} {us:7.1f} | {b['mfma'] * 1e6:6.1f} {b['hbm'] * 1e6:6.1f} {b['feed'] * 1e6:6.1f} {b['lds'] * 1e6:6.1f} | "
              f"{which:>5s} {us / (bound * 1e6):10.2f}  {tile}")
    print(f"\nsum over the step: measured {tot_meas / 1e3:.2f} ms, sum of per-launch bounds {tot_bound / 1e3:.2f} ms")
    for k, (m, bd) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"  bound by {k:5s}: measured {m / 1e3:6.2f} ms vs bound {bd / 1e3:6.2f} ms")


if __name__ == "__main__":
    main()
