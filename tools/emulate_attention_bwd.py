#!/usr/bin/env python
"""Lane-level CPU emulation of storygen_amd/csrc/attention_bwd.hip (development aid: the kernel was written without access
to a GPU, so its index arithmetic — LDS-DMA source coordinates, XOR swizzles, the bit-permuted A rows, the
register-resident dS / P feeding the second contraction, the output mapping — is transliterated here statement by
statement on top of a model of v_mfma_f32_32x32x16_f16 and checked against oracle.storygen_backward.attention_core_bwd).

MFMA model — the operand layout the hardware-validated forward kernel is built on (tests/test_kernels_gpu.py::test_mfma_fragment_layout pins it on
the device) and the C/D map of the CDNA4 guide: D[i][j] += sum_k A[i][k] B[k][j];
A fragment of lane l = A[l & 31][8 (l >> 5) + 0..7], B fragment = B[8 (l >> 5) + 0..7][l & 31],
accumulator register r of lane l = D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
LDS starts as NaN, so a read of a byte no DMA wrote (or of a row beyond the tile that is not multiplied by a zero owned
fragment) poisons the result exactly as it would on hardware.

    python tools/emulate_attention_bwd.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import storygen_backward as B  # noqa: E402


def mfma(a, b, acc):
    """a, b: [64 lanes][8]; acc: [64 lanes][16] -> new acc."""
    A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = a[l]
        Bm[8 * (l >> 5): 8 * (l >> 5) + 8, l & 31] = b[l]
    Dm = A @ Bm
    out = acc.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += Dm[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def kswz(D, row):
    return 0 if D == 40 else ((row >> 3) & 1 if D == 80 else (row >> 2) & 3)


def f16(x):
    return np.asarray(x, dtype=np.float16).astype(np.float64)


def emulate(D, NW, dkv, q, k, v, do, lse2, delta, scale):
    """One (batch, head).  q, do: [Nq, D]; k, v: [Nk, D] (float64 holding fp16-exact values).  Returns dq [Nq, D] or
    (dkT [D, Nk], dvT [D, Nk])."""
    Nq, Nk = q.shape[0], k.shape[0]
    DC, NDK, DT, ROW = D // 8, (D + 15) // 16, (D + 31) // 32, D * 2
    TOK_BYTES, TR_BYTES = 64 * ROW, D * 128
    TOK_SEG, TR_SEG = TOK_BYTES // 1024, TR_BYTES // 1024
    NTR = 2 if dkv else 1
    OFF_T2, OFF_T3 = TOK_BYTES, 2 * TOK_BYTES
    OFF_T4, OFF_LD = OFF_T3 + TR_BYTES, OFF_T3 + NTR * TR_BYTES
    STAGE = OFF_LD + (1024 if dkv else 0)
    n_own, n_str = (Nk, Nq) if dkv else (Nq, Nk)
    nstr8 = (n_str + 7) & ~7
    X1, X2 = (q, do) if dkv else (k, v)                      # streamed token-major [n_str, D]
    pad = lambda m: np.pad(m, ((0, 0), (0, nstr8 + 64 - m.shape[1])))     # noqa: E731  (finite padding of the transposed rows)
    X1T, X2T = pad(X1.T.copy()), pad(do.T.copy())            # [D, n_str padded]
    Y1, Y2 = (k, v) if dkv else (q, do)                      # owned
    LD = np.stack([lse2, delta], 1).reshape(-1)              # [Nq * 2]
    scale_log2 = scale * 1.4426950408889634
    nob = -(-n_own // (32 * NW))
    out1 = np.zeros((D, n_own)) if dkv else np.zeros((n_own, D))
    out2 = np.zeros((D, n_own))
    for ob in range(nob):
        lds = np.full(STAGE // 2 + 64, np.nan)               # halves; the LD region is addressed as floats via a side array
        ldsf = np.full(256, np.nan)                          # 1 KiB of floats for the (lse2, delta) pairs
        state = []
        for wave in range(NW):
            own0 = (ob * NW + wave) * 32
            lanes = np.arange(64)
            l31, hi = lanes & 31, lanes >> 5
            oi = np.minimum(own0 + l31, n_own - 1)
            f1 = np.zeros((NDK, 64, 8)); f2 = np.zeros((NDK, 64, 8))
            for s in range(NDK):
                for l in range(64):
                    d0 = s * 16 + hi[l] * 8
                    if d0 < D:
                        f1[s, l] = Y1[oi[l], d0:d0 + 8]; f2[s, l] = Y2[oi[l], d0:d0 + 8]
            own_lse = LD[oi * 2] if not dkv else None
            own_dl = LD[oi * 2 + 1] if not dkv else None
            state.append(dict(own0=own0, f1=f1, f2=f2, own_lse=own_lse, own_dl=own_dl,
                              acc1=np.zeros((DT, 64, 16)), acc2=np.zeros((DT, 64, 16))))
        ntiles = -(-n_str // 64)
        for tile in range(ntiles):
            s0 = tile * 64
            full = s0 + 64 <= n_str
            # ---- issue (all waves): region copies exactly as issue_region computes them
            for wave in range(NW):
                for region, (src, NS, base) in enumerate([(X1, TOK_SEG, 0), (X2, TOK_SEG, OFF_T2), (X1T, TR_SEG, OFF_T3)]
                                                         + ([(X2T, TR_SEG, OFF_T4)] if dkv else [])):
                    for j in range(-(-NS // NW)):
                        gl = j * NW + wave
                        if gl >= NS:
                            continue
                        for lane in range(64):
                            s = gl * 64 + lane
                            if region < 2:
                                row = s // DC
                                col = ((s - row * DC) ^ kswz(D, row)) * 8
                                r_ = s0 + row if full else min(s0 + row, n_str - 1)
                                vals = src[r_, col:col + 8]
                            else:
                                row = s >> 3
                                col = ((s & 7) ^ ((row >> 1) & 7)) * 8
                                c_ = s0 + col if full else min(s0 + col, nstr8 - 8)
                                vals = src[row, c_:c_ + 8]
                            dst = (base + gl * 1024 + lane * 16) // 2
                            lds[dst:dst + 8] = vals
                if dkv and wave == NW - 1:
                    for lane in range(64):
                        f = min(s0 * 2 + lane * 4, n_str * 2 - 4)
                        ldsf[lane * 4: lane * 4 + 4] = LD[f:f + 4]
            # ---- compute per wave
            for st_ in state:
                lanes = np.arange(64)
                l31, hi = lanes & 31, lanes >> 5
                prow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1)
                sT = np.zeros((2, 64, 16)); pT = np.zeros((2, 64, 16))
                for kb in range(2):
                    for s in range(NDK):
                        a1 = np.zeros((64, 8)); a2 = np.zeros((64, 8))
                        for l in range(64):
                            row = kb * 32 + prow[l]
                            o = row * ROW + (((s * 2 + hi[l]) ^ kswz(D, row)) << 4)
                            a1[l] = lds[o // 2: o // 2 + 8]
                            a2[l] = lds[(OFF_T2 + o) // 2: (OFF_T2 + o) // 2 + 8]
                        sT[kb] = mfma(a1, st_["f1"][s], sT[kb] if s else np.zeros((64, 16)))
                        pT[kb] = mfma(a2, st_["f2"][s], pT[kb] if s else np.zeros((64, 16)))
                for kb in range(2):
                    for g in range(2):
                        for l in range(64):
                            first = kb * 32 + 16 * g + 8 * hi[l]
                            for j in range(8):
                                if dkv:
                                    lse, dl = ldsf[(first + j) * 2], ldsf[(first + j) * 2 + 1]
                                else:
                                    lse, dl = st_["own_lse"][l], st_["own_dl"][l]
                                r = 8 * g + j
                                valid = tile * 64 + first + j < n_str
                                pv = 2.0 ** (sT[kb, l, r] * scale_log2 - lse) if valid else 0.0
                                sT[kb, l, r] = pv
                                pT[kb, l, r] = pv * (pT[kb, l, r] - dl) if valid else 0.0
                for ks in range(4):
                    dsf = f16(pT[ks >> 1][:, (ks & 1) * 8:(ks & 1) * 8 + 8])
                    pf = f16(sT[ks >> 1][:, (ks & 1) * 8:(ks & 1) * 8 + 8])
                    for i in range(DT):
                        a3 = np.zeros((64, 8)); a4 = np.zeros((64, 8))
                        for l in range(64):
                            d = min(i * 32 + l31[l], D - 1)
                            o = d * 128 + (((ks * 2 + hi[l]) ^ ((d >> 1) & 7)) << 4)
                            a3[l] = lds[(OFF_T3 + o) // 2: (OFF_T3 + o) // 2 + 8]
                            if dkv:
                                a4[l] = lds[(OFF_T4 + o) // 2: (OFF_T4 + o) // 2 + 8]
                        st_["acc1"][i] = mfma(a3, dsf, st_["acc1"][i])
                        if dkv:
                            st_["acc2"][i] = mfma(a4, pf, st_["acc2"][i])
        # ---- store
        for st_ in state:
            for l in range(64):
                orow = st_["own0"] + (l & 31)
                if orow >= n_own:
                    continue
                for i in range(DT):
                    for r in range(16):
                        d = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                        if d < D:
                            if dkv:
                                out1[d, orow] = st_["acc1"][i][l, r] * scale
                                out2[d, orow] = st_["acc2"][i][l, r]
                            else:
                                out1[orow, d] = st_["acc1"][i][l, r] * scale
    return (out1, out2) if dkv else out1


def main():
    rng = np.random.default_rng(0)
    for D, Nq, Nk, NW in ((40, 64, 104, 1), (40, 72, 77, 2), (80, 64, 96, 1), (160, 40, 72, 1)):
        q, k, v, do = (f16(rng.standard_normal((n, D))) for n in (Nq, Nk, Nk, Nq))
        scale = D ** -0.5
        tq, tk, tv, tdo = (torch.tensor(x)[None] for x in (q, k, v, do))
        o, lse = B.attention_core(tq, tk, tv, 1)
        dq_ref, dk_ref, dv_ref = (x[0].numpy() for x in B.attention_core_bwd(tq, tk, tv, o, lse, tdo, 1))
        lse2 = lse[0, 0].numpy() * 1.4426950408889634
        delta = (do * f16(o[0].numpy())).sum(1)
        rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)      # noqa: E731
        dq = emulate(D, NW, False, q, k, v, do, lse2, delta, scale)
        e_q = rel(dq, dq_ref)
        msg = f"D={D} Nq={Nq} Nk={Nk} NW={NW}: dq {e_q:.1e}"
        if Nq % 8 == 0:
            dkt, dvt = emulate(D, NW, True, q, k, v, do, lse2, delta, scale)
            e_k, e_v = rel(dkt.T, dk_ref), rel(dvt.T, dv_ref)
            msg += f"  dk {e_k:.1e}  dv {e_v:.1e}"
            assert e_k < 5e-3 and e_v < 5e-3, msg
        print(msg)
        assert e_q < 5e-3, msg
    print("EMULATION_OK")


if __name__ == "__main__":
    main()
