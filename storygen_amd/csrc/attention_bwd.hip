// Attention backward for the stage-2 training step (BASELINE config 4: "fwd+bwd of HIP attention"), gfx950.
// STATUS: validated on MI355X in round 2 (tests/test_backward_gpu.py: D = 40 / 80 / 160, partial tiles, 4096 x 4096).
// Formulas: oracle/storygen_backward.py::attention_core_bwd (checked against torch.autograd on the CPU):
//   P = exp2(scale_log2 * S - lse2)          lse2 = the forward pass's log2-domain log-sum-exp row (sg_attn_fwd_lse_f16)
//   delta[q] = sum_d dO[q,d] O[q,d]          (sg_attn_bwd_prep_f32 packs (lse2, delta) pairs)
//   dV = P^T dO,  dP = dO V^T,  dS = P * (dP - delta),  dQ = scale * dS K,  dK = scale * dS^T Q
//
// Two launches of ONE kernel template, no atomics, deterministic:
//   DKV = false (dQ):     a workgroup OWNS 32 queries per wave and STREAMS the keys in tiles of 64;
//   DKV = true (dK, dV):  a workgroup OWNS 32 keys per wave and STREAMS the queries in tiles of 64.
// Both recompute S and dP for every (owned, streamed) pair with the forward kernel's MFMA formulation
// (v_mfma_f32_32x32x16_f16): the streamed operand is the A matrix, read from LDS with its rows bit-permuted so that the 8
// accumulator registers of a 16-row step are 8 CONSECUTIVE streamed rows; the owned operand is the B matrix, held in
// registers.  The accumulator of lane (owned row l & 31, hi = l >> 5) therefore holds, in register r, streamed row
// 16 (r >> 3) + 8 hi + (r & 7) — which is exactly the B-operand layout of the second contraction
//   acc^T[d, owned] += X^T[d, streamed] . dS^T[streamed, owned]      (X^T = K^T for dQ, Q^T for dK, dO^T with P for dV),
// so P / dS never move across lanes or through LDS (same trick as the forward kernel's O^T = V^T P^T).
// The streamed side is needed both token-major ([rows][D], A operand of S and dP) and transposed ([D][rows], A operand of
// the accumulation); like V^T in the forward pass the transposed copies come from the host for free (a projection GEMM
// with swapped operands), so every LDS tile is a plain 16-byte-chunk copy of global memory filled by LDS-DMA.
//   streamed tiles per 64 rows:  dQ : K [64][D], V [64][D], K^T [D][64]
//                                dKV: Q [64][D], dO [64][D], Q^T [D][64], dO^T [D][64], 64 (lse2, delta) pairs
// Per-row scalars: for dQ they belong to the OWNED row (one pair per lane); for dKV to the STREAMED rows (16 pairs per
// lane and 32-row block, read from the LDS tile).
#include "attention_kernel.h"
#include <type_traits>

namespace sgattn {

struct AttnBwdParams {
    const f16* q;    long ldq, bsq;
    const f16* qt;   long ldqt, bsqt;
    const f16* k;    long ldk, bsk;
    const f16* kt;   long ldkt, bskt;
    const f16* v;    long ldv, bsv;
    const f16* dout; long lddo, bsdo;
    const f16* dot;  long lddot, bsdot;
    const float* ld2;                       // [B, H, Nq][2] = (lse2, delta)
    f16* dq;  long lddq, bsdq;
    f16* dkt; long lddkt, bsdkt;
    f16* dvt; long lddvt, bsdvt;
    int B, H, Nq, Nk, nob;                  // nob = owned-row blocks (32 * NW rows) per (batch, head)
    float scale_log2, scale;
};

// ACC (dKV only): 3 = accumulate dK and dV in one pass; 1 = dK only, 2 = dV only — at D = 160 both accumulators (160
// registers) plus the owned fragments (80) do not fit, so that head dim runs the pass twice (the 16x16 level is tiny).
#ifdef SG_BWD_DKV_OCC2          // A/B build (tools/build_variant.py): the D = 40 dK/dV pass held to 256 registers = two waves per SIMD
#define SG_BWD_WAVES(D, DKV) __attribute__((amdgpu_waves_per_eu((D) == 40 && (DKV) ? 2 : 1, 2)))
#else
#define SG_BWD_WAVES(D, DKV)
#endif
template <int D, int NW, int S, bool DKV, int ACC = 3>
__global__ __launch_bounds__(64 * NW) SG_BWD_WAVES(D, DKV) void attn_bwd_kernel(const AttnBwdParams p) {
    constexpr bool DO1 = !DKV || (ACC & 1), DO2 = DKV && (ACC & 2);
    constexpr int DC = D / 8, NDK = (D + 15) / 16, DT = (D + 31) / 32, ROW = D * 2;
    constexpr int TOK_BYTES = 64 * ROW, TR_BYTES = D * 128;
    constexpr int TOK_SEG = TOK_BYTES / 1024, TR_SEG = TR_BYTES / 1024;       // both D / 8
    constexpr int NTR = DKV ? 2 : 1, LD_SEG = DKV ? 1 : 0;
    constexpr int OFF_T2 = TOK_BYTES, OFF_T3 = 2 * TOK_BYTES, OFF_T4 = OFF_T3 + TR_BYTES, OFF_LD = OFF_T3 + NTR * TR_BYTES;
    constexpr int STAGE = OFF_LD + LD_SEG * 1024;
    static_assert(S >= 1 && S <= 3, "1 .. 3 stages");
    static_assert(S * STAGE + 16 <= 160 * 1024, "ring must fit the LDS");
    __shared__ __attribute__((aligned(16))) char smem[S * STAGE + 16];

    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = work / p.nob, ob = work - bh * p.nob;
    const int h = bh / p.B, b = bh - h * p.B;
    const int own0 = (ob * NW + wave) * 32;
    const int n_own = DKV ? p.Nk : p.Nq, n_str = DKV ? p.Nq : p.Nk;
    const int nstr8 = (n_str + 7) & ~7;

    // streamed operands: X1 (A of S), X2 (A of dP) token-major; X1T (and X2T) transposed
    const f16* X1 = (DKV ? p.q + (long)b * p.bsq : p.k + (long)b * p.bsk) + (long)h * D;
    const f16* X2 = (DKV ? p.dout + (long)b * p.bsdo : p.v + (long)b * p.bsv) + (long)h * D;
    const long ld1 = DKV ? p.ldq : p.ldk, ld2_ = DKV ? p.lddo : p.ldv;
    const long ldt1 = DKV ? p.ldqt : p.ldkt, ldt2 = p.lddot;
    const f16* X1T = (DKV ? p.qt + (long)b * p.bsqt : p.kt + (long)b * p.bskt) + (long)h * D * ldt1;
    const f16* X2T = DKV ? p.dot + (long)b * p.bsdot + (long)h * D * ldt2 : nullptr;
    const float* LD = p.ld2 + ((long)b * p.H + h) * p.Nq * 2;
    // owned operands (B matrices): Y1 pairs with X1 in S, Y2 pairs with X2 in dP
    const f16* Y1 = (DKV ? p.k + (long)b * p.bsk : p.q + (long)b * p.bsq) + (long)h * D;
    const f16* Y2 = (DKV ? p.v + (long)b * p.bsv : p.dout + (long)b * p.bsdo) + (long)h * D;
    const long ldy1 = DKV ? p.ldk : p.ldq, ldy2 = DKV ? p.ldv : p.lddo;

    // ---- LDS-DMA issue.  A tile consists of up to five regions (0: X1 rows, 1: X2 rows, 2: X1T, 3: X2T, 4: the (lse2, delta)
    // pairs), each a whole number of 1 KiB segments (= one wave instruction); wave w issues segments w, w + NW, ... of every
    // region, so the region — and with it the base pointer and stride — is a compile-time property of each instruction.
    // FULL tiles (all 64 streamed rows exist — every tile but possibly the last): the per-lane part of a piece's source address does not
    // depend on the tile, it is computed ONCE here as a 32-bit element offset; per tile only the wave-uniform base moves (round 6: the
    // per-tile row / column / clamp arithmetic was ~60 VALU instructions per tile and wave, issued beside the MFMAs where they add).
    const int ntiles = (n_str + 63) / 64, nfull = n_str / 64;
    constexpr int TOK_IT = (TOK_SEG + NW - 1) / NW, TR_IT = (TR_SEG + NW - 1) / NW;
    unsigned tk1[TOK_IT], tk2[TOK_IT], tr1[TR_IT], tr2[TR_IT];
#pragma unroll
    for (int j = 0; j < TOK_IT; ++j) {
        const int s = (j * NW + wave) * 64 + lane, row = s / DC;
        const int col = ((s - row * DC) ^ kswz<D>(row)) * 8;
        tk1[j] = (unsigned)(row * (int)ld1 + col);            // < 64 * ld (validated on the host)
        tk2[j] = (unsigned)(row * (int)ld2_ + col);
    }
#pragma unroll
    for (int j = 0; j < TR_IT; ++j) {
        const int s = (j * NW + wave) * 64 + lane, row = s >> 3;
        const int col = ((s & 7) ^ ((row >> 1) & 7)) * 8;
        tr1[j] = (unsigned)(row * (int)ldt1 + col);           // < D * ldt (validated on the host)
        tr2[j] = DKV ? (unsigned)(row * (int)ldt2 + col) : 0u;
    }
    auto issue_region = [&](auto R, const f16* base_ptr, long ld, char* lds_region, int s0) __attribute__((always_inline)) {
        constexpr int REGION = decltype(R)::value;             // (partial tile: rows / columns clamped to the last valid one)
        constexpr int NS = REGION < 2 ? TOK_SEG : TR_SEG;
#pragma unroll
        for (int j = 0; j < (NS + NW - 1) / NW; ++j) {
            const int gl = j * NW + wave;                       // wave-uniform
            if (gl < NS) {
                const int s = gl * 64 + lane;
                const f16* src;
                if constexpr (REGION < 2) {                      // token-major rows [64][D]
                    const int row = s / DC;
                    const int col = ((s - row * DC) ^ kswz<D>(row)) * 8;
                    src = base_ptr + (long)min(s0 + row, n_str - 1) * ld + col;
                } else {                                         // transposed [D][64]
                    const int row = s >> 3;
                    const int col = ((s & 7) ^ ((row >> 1) & 7)) * 8;
                    src = base_ptr + (long)row * ld + min(s0 + col, nstr8 - 8);
                }
                glds16(src, lds_region + gl * 1024);
            }
        }
    };
    auto issue = [&](int tile, auto stage_c, auto full_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_c)::value;
        constexpr bool FULL = decltype(full_c)::value;
        char* base = smem + ST * STAGE;
        const int s0 = tile * 64;
        if constexpr (FULL) {
            const f16* b1 = X1 + (long)s0 * ld1;
            const f16* b2 = X2 + (long)s0 * ld2_;
#pragma unroll
            for (int j = 0; j < TOK_IT; ++j) {
                const int gl = j * NW + wave;
                if (gl < TOK_SEG) {
                    glds16(b1 + tk1[j], base + gl * 1024);
                    glds16(b2 + tk2[j], base + OFF_T2 + gl * 1024);
                }
            }
            const f16* b3 = X1T + s0;
            const f16* b4 = DKV ? X2T + s0 : nullptr;
#pragma unroll
            for (int j = 0; j < TR_IT; ++j) {
                const int gl = j * NW + wave;
                if (gl < TR_SEG) {
                    glds16(b3 + tr1[j], base + OFF_T3 + gl * 1024);
                    if constexpr (DKV) glds16(b4 + tr2[j], base + OFF_T4 + gl * 1024);
                }
            }
        } else {
            issue_region(std::integral_constant<int, 0>{}, X1, ld1, base, s0);
            issue_region(std::integral_constant<int, 1>{}, X2, ld2_, base + OFF_T2, s0);
            issue_region(std::integral_constant<int, 2>{}, X1T, ldt1, base + OFF_T3, s0);
            if constexpr (DKV) issue_region(std::integral_constant<int, 3>{}, X2T, ldt2, base + OFF_T4, s0);
        }
        if constexpr (DKV) {
            if (wave == NW - 1) {   // the pairs: 2 floats per streamed row; 16-byte chunks clamped to the last valid one (masked)
                const int f = min(s0 * 2 + lane * 4, n_str * 2 - 4);
                glds16(reinterpret_cast<const f16*>(LD + f), base + OFF_LD);
            }
        }
    };
    // the tile after `tile` into stage ST (full or partial, whichever it is)
    auto issue_next = [&](int tile, auto stage_c) __attribute__((always_inline)) {
        if (tile < nfull) issue(tile, stage_c, std::true_type{});
        else if (tile < ntiles) issue(tile, stage_c, std::false_type{});
    };

    // ---- owned fragments (B operands): lane = (owned row l31, d-chunk 2s + hi); rows beyond n_own clamped (never stored)
    f16x8 f1[NDK], f2[NDK];
    const int oi = min(own0 + l31, n_own - 1);
#pragma unroll
    for (int s = 0; s < NDK; ++s) {
        const int d0 = s * 16 + hi * 8;
        H8 a, c; a.u = make_uint4(0, 0, 0, 0); c.u = a.u;
        if (d0 < D) { a.u = ldg16(Y1 + (long)oi * ldy1 + d0); c.u = ldg16(Y2 + (long)oi * ldy2 + d0); }
        f1[s] = a.v; f2[s] = c.v;
    }
    float own_lse = 0.f, own_delta = 0.f;
    if constexpr (!DKV) {
        const float2 pr = *reinterpret_cast<const float2*>(LD + (long)oi * 2);
        own_lse = pr.x; own_delta = pr.y;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): inside the loop the only vector-memory traffic is the LDS-DMA ring

    f32x16 acc1[DO1 ? DT : 1], acc2[DO2 ? DT : 1];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (DO1) acc1[i][r] = 0.f;
            if constexpr (DO2) acc2[i][r] = 0.f;
        }

    const int prow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);          // pi(l31): swap bits 2 and 3
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    if constexpr (S >= 2) issue_next(0, C0{});
    if constexpr (S == 3) issue_next(1, C1{});
    // S = 3 (round 6): tile t + 1 stays in flight while tile t is waited for — a counted wait.  The DMA pieces a wave issues per tile
    // are the same for every tile but differ between the waves (wave 0 takes the odd segments, the last wave the (lse2, delta) pairs).
    constexpr int PW_REST = (TOK_SEG / NW) * 2 + (TR_SEG / NW) * NTR;                                   // waves without extras
    auto wait_landed = [&](int tile) __attribute__((always_inline)) {
        if constexpr (S == 3) {
            if (tile + 1 < ntiles) {
                const int extra = ((wave < TOK_SEG % NW) ? 2 : 0) + ((wave < TR_SEG % NW) ? NTR : 0) + ((DKV && wave == NW - 1) ? 1 : 0);
                // (at most 2 + NTR + 1 extras: one branch per count, each with its immediate)
                switch (extra) {
                    case 0: wait_vmcnt<PW_REST>(); break;
                    case 1: wait_vmcnt<PW_REST + 1>(); break;
                    case 2: wait_vmcnt<PW_REST + 2>(); break;
                    case 3: wait_vmcnt<PW_REST + 3>(); break;
                    case 4: wait_vmcnt<PW_REST + 4>(); break;
                    default: wait_vmcnt<PW_REST + 5>(); break;
                }
                return;
            }
        }
        wait_vmcnt<0>();
    };
    // One streamed tile.  The ring stage and "all 64 rows exist" are compile-time properties of each copy of the body (round 6): every
    // LDS address is a loop-invariant per-lane base + an immediate, and the full tiles carry no row masks.
    auto body = [&](int tile, auto stage_c, auto full_c) __attribute__((always_inline)) {
        constexpr int ST = decltype(stage_c)::value;
        constexpr bool FULL = decltype(full_c)::value;
        if constexpr (S == 1) issue(tile, stage_c, full_c);
        wait_landed(tile);
        __builtin_amdgcn_s_barrier();
        if constexpr (S >= 2) issue_next(tile + S - 1, std::integral_constant<int, (ST + S - 1) % S>{});
        const char* T1 = smem + ST * STAGE;
        const char* T2 = T1 + OFF_T2;
        const char* T3 = T1 + OFF_T3;
        const char* T4 = T1 + OFF_T4;
        const char* TL = T1 + OFF_LD;

        // ---- S^T and dP^T of the two 32-row blocks: lane = owned row, register r = streamed row 16(r>>3) + 8hi + (r&7)
        f32x16 sT[2], pT[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int row = kb * 32 + prow;
            const int sw = kswz<D>(row);
#pragma unroll
            for (int st = 0; st < NDK; ++st) {
                const int o = row * ROW + (((st * 2 + hi) ^ sw) << 4);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(T1 + o);
                const f16x8 a2 = *reinterpret_cast<const f16x8*>(T2 + o);
                sT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, f1[st], st == 0 ? zero16 : sT[kb], 0, 0, 0);
                pT[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, f2[st], st == 0 ? zero16 : pT[kb], 0, 0, 0);
            }
        }
        // ---- P (into sT) and dS (into pT); streamed rows beyond n_str contribute nothing
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int first = kb * 32 + 16 * g + 8 * hi;             // streamed row of register 8g within the tile
                float lse[8], dl[8];
                if constexpr (DKV) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float4 v4 = *reinterpret_cast<const float4*>(TL + (first + 2 * m) * 8);
                        lse[2 * m] = v4.x; dl[2 * m] = v4.y; lse[2 * m + 1] = v4.z; dl[2 * m + 1] = v4.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { lse[j] = own_lse; dl[j] = own_delta; }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 8 * g + j;
                    const bool valid = FULL || tile * 64 + first + j < n_str;
                    const float pv = valid ? __builtin_amdgcn_exp2f(fmaf(sT[kb][r], p.scale_log2, -lse[j])) : 0.f;
                    sT[kb][r] = pv;
                    pT[kb][r] = valid ? pv * (pT[kb][r] - dl[j]) : 0.f;
                }
            }
        // ---- acc1^T += X1^T dS^T  (and acc2^T += X2^T P^T): 4 k-steps of 16 streamed rows, B = this lane's own registers
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 dsf, pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dsf[j] = (f16)pT[ks >> 1][(ks & 1) * 8 + j];
                pf[j] = (f16)sT[ks >> 1][(ks & 1) * 8 + j];
            }
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const int d = min(i * 32 + l31, D - 1);      // rows >= D: duplicates, never stored
                const int o = d * 128 + (((ks * 2 + hi) ^ ((d >> 1) & 7)) << 4);
                if constexpr (DO1)
                    acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(T3 + o), dsf, acc1[i], 0, 0, 0);
                if constexpr (DO2)
                    acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(T4 + o), pf, acc2[i], 0, 0, 0);
            }
        }
        if constexpr (S == 1) __builtin_amdgcn_s_barrier();   // every wave is done with the only stage before it is refilled
    };
    if constexpr (S == 3) {
        for (int tile = 0; tile < nfull; tile += 3) {
            body(tile, C0{}, std::true_type{});
            if (tile + 1 >= nfull) break;
            body(tile + 1, C1{}, std::true_type{});
            if (tile + 2 >= nfull) break;
            body(tile + 2, C2{}, std::true_type{});
        }
        if (nfull < ntiles) {                                  // the partial tile sits in stage nfull % 3
            const int st = nfull % 3;
            if (st == 0) body(nfull, C0{}, std::false_type{});
            else if (st == 1) body(nfull, C1{}, std::false_type{});
            else body(nfull, C2{}, std::false_type{});
        }
    } else if constexpr (S == 2) {
        for (int tile = 0; tile < nfull; tile += 2) {
            body(tile, C0{}, std::true_type{});
            if (tile + 1 >= nfull) break;
            body(tile + 1, C1{}, std::true_type{});
        }
        if (nfull < ntiles) {                                  // the partial tile sits in stage nfull & 1
            if (nfull & 1) body(nfull, C1{}, std::false_type{});
            else body(nfull, C0{}, std::false_type{});
        }
    } else {
        for (int tile = 0; tile < nfull; ++tile) body(tile, C0{}, std::true_type{});
        if (nfull < ntiles) body(nfull, C0{}, std::false_type{});
    }

    // ---- store (lane holds d = 32i + (r&3) + 8(r>>2) + 4hi of its owned row)
    const int orow = own0 + l31;
    if (orow >= n_own) return;
    if constexpr (!DKV) {
        f16* O = p.dq + (long)b * p.bsdq + (long)orow * p.lddq + (long)h * D;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = i * 32 + 8 * g + 4 * hi;
                if (d0 < D) {
                    f16x4 w = {(f16)(acc1[i][4 * g + 0] * p.scale), (f16)(acc1[i][4 * g + 1] * p.scale),
                               (f16)(acc1[i][4 * g + 2] * p.scale), (f16)(acc1[i][4 * g + 3] * p.scale)};
                    *reinterpret_cast<f16x4*>(O + d0) = w;
                }
            }
    } else {
        f16* OK_ = p.dkt + (long)b * p.bsdkt + (long)h * D * p.lddkt + orow;
        f16* OV_ = p.dvt + (long)b * p.bsdvt + (long)h * D * p.lddvt + orow;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (d < D) {
                    if constexpr (DO1) OK_[(long)d * p.lddkt] = (f16)(acc1[i][r] * p.scale);
                    if constexpr (DO2) OV_[(long)d * p.lddvt] = (f16)acc2[i][r];
                }
            }
    }
}

// (lse2, delta) pairs: delta[b,h,q] = sum_d dO[b,q,h*D+d] * O[b,q,h*D+d]; one thread per (b, h, q)
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const f16* o, long ldo, long bso, const f16* dout, long lddo, long bsdo,
                                                            const float* lse2, float* ld2, int B, int H, int Nq, int D) {
    const long total = (long)B * H * Nq;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int q = (int)(idx % Nq);
        const int bh = (int)(idx / Nq), h = bh % H, b = bh / H;
        const f16* po = o + (long)b * bso + (long)q * ldo + (long)h * D;
        const f16* pd = dout + (long)b * bsdo + (long)q * lddo + (long)h * D;
        float acc = 0.f;
        for (int d = 0; d < D; d += 8) {
            H8 a, c; a.u = ldg16(po + d); c.u = ldg16(pd + d);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)a.h[j] * (float)c.h[j];
        }
        *reinterpret_cast<float2*>(ld2 + idx * 2) = make_float2(lse2[idx], acc);
    }
}

template <int D, int NW, int S, bool DKV, int ACC = 3>
void launch_bwd(const AttnBwdParams& p0, hipStream_t st) {
    AttnBwdParams p = p0;
    p.nob = sg_cdiv(DKV ? p.Nk : p.Nq, 32 * NW);
    hipLaunchKernelGGL((attn_bwd_kernel<D, NW, S, DKV, ACC>), dim3(p.nob * p.H * p.B), dim3(64 * NW), 0, st, p);
}

}  // namespace sgattn

using namespace sgattn;

#ifndef SG_BWD_RING
#define SG_BWD_RING 2          // ring depth of the D = 40 passes and the D = 80 dK/dV pass (A/B builds: tools/build_variant.py ... -DSG_BWD_RING=3)
#endif

static int check_bwd_desc(const sg_attn_bwd_desc* d, bool dkv, const char* who) {
    SG_REQUIRE(d != nullptr, "%s: null descriptor", who);
    SG_REQUIRE(d->q && d->k && d->v && d->dout && d->ld2, "%s: null q/k/v/dout/ld2", who);
    SG_REQUIRE(d->B > 0 && d->H > 0 && d->Nq > 0 && d->Nk > 0, "%s: bad shape", who);
    if (d->D != 40 && d->D != 80 && d->D != 160) return sg_set_error(SG_EUNSUP, "%s: head dim %d not in {40, 80, 160}", who, d->D);
    // dkv streams the queries: their (lse2, delta) pairs are fetched 16 bytes at a time and Q^T / dO^T rows must hold
    // finite data up to Nq rounded up to 8.  dq streams the keys and accepts any Nk (text attention: 77) as long as the
    // K^T rows are finite up to Nk rounded up to 8 (ldkt >= that).
    SG_REQUIRE(!dkv || d->Nq % 8 == 0, "%s: Nq=%d must be a multiple of 8", who, d->Nq);
    const int64_t hd = (int64_t)d->H * d->D;
    SG_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->lddo % 8 == 0 && d->ldq >= hd && d->ldk >= hd &&
                   d->ldv >= hd && d->lddo >= hd, "%s: token strides", who);
    SG_REQUIRE(d->bsq % 8 == 0 && d->bsk % 8 == 0 && d->bsv % 8 == 0 && d->bsdo % 8 == 0, "%s: batch strides", who);
    SG_REQUIRE(sg_aligned16(d->q) && sg_aligned16(d->k) && sg_aligned16(d->v) && sg_aligned16(d->dout) && sg_aligned16(d->ld2),
               "%s: 16-byte alignment", who);
    if (dkv) {
        SG_REQUIRE(d->qt && d->dot && d->dkt && d->dvt, "%s: null qt/dot/dkt/dvt", who);
        SG_REQUIRE(d->ldqt % 8 == 0 && d->lddot % 8 == 0 && d->ldqt >= d->Nq && d->lddot >= d->Nq && d->bsqt % 8 == 0 &&
                       d->bsdot % 8 == 0 && sg_aligned16(d->qt) && sg_aligned16(d->dot), "%s: transposed inputs", who);
        SG_REQUIRE(d->lddkt >= d->Nk && d->lddvt >= d->Nk, "%s: transposed outputs", who);
        SG_REQUIRE((int64_t)d->D * d->ldqt < (1ll << 31) && (int64_t)d->D * d->lddot < (1ll << 31), "%s: 32-bit offsets", who);
    } else {
        SG_REQUIRE(d->kt && d->dq, "%s: null kt/dq", who);
        SG_REQUIRE(d->ldkt % 8 == 0 && d->ldkt >= ((d->Nk + 7) & ~7) && d->bskt % 8 == 0 && sg_aligned16(d->kt),
                   "%s: transposed input (ldkt must cover Nk rounded up to 8)", who);
        SG_REQUIRE(d->lddq % 4 == 0 && d->bsdq % 4 == 0 && d->lddq >= hd, "%s: dq strides", who);
        SG_REQUIRE((int64_t)d->D * d->ldkt < (1ll << 31), "%s: 32-bit offsets", who);
    }
    SG_REQUIRE((int64_t)64 * d->ldq < (1ll << 31) && (int64_t)64 * d->ldk < (1ll << 31) && (int64_t)64 * d->ldv < (1ll << 31) &&
                   (int64_t)64 * d->lddo < (1ll << 31), "%s: 32-bit tile offsets", who);
    return SG_OK;
}

static AttnBwdParams bwd_params(const sg_attn_bwd_desc* d) {
    AttnBwdParams p{};
    p.q = reinterpret_cast<const f16*>(d->q); p.ldq = d->ldq; p.bsq = d->bsq;
    p.qt = reinterpret_cast<const f16*>(d->qt); p.ldqt = d->ldqt; p.bsqt = d->bsqt;
    p.k = reinterpret_cast<const f16*>(d->k); p.ldk = d->ldk; p.bsk = d->bsk;
    p.kt = reinterpret_cast<const f16*>(d->kt); p.ldkt = d->ldkt; p.bskt = d->bskt;
    p.v = reinterpret_cast<const f16*>(d->v); p.ldv = d->ldv; p.bsv = d->bsv;
    p.dout = reinterpret_cast<const f16*>(d->dout); p.lddo = d->lddo; p.bsdo = d->bsdo;
    p.dot = reinterpret_cast<const f16*>(d->dot); p.lddot = d->lddot; p.bsdot = d->bsdot;
    p.ld2 = d->ld2;
    p.dq = reinterpret_cast<f16*>(d->dq); p.lddq = d->lddq; p.bsdq = d->bsdq;
    p.dkt = reinterpret_cast<f16*>(d->dkt); p.lddkt = d->lddkt; p.bsdkt = d->bsdkt;
    p.dvt = reinterpret_cast<f16*>(d->dvt); p.lddvt = d->lddvt; p.bsdvt = d->bsdvt;
    p.B = d->B; p.H = d->H; p.Nq = d->Nq; p.Nk = d->Nk;
    p.scale = d->scale; p.scale_log2 = d->scale * 1.44269504088896340736f;
    return p;
}

extern "C" int sg_attn_bwd_dq_f16(const sg_attn_bwd_desc* d, sg_stream_t stream) {
    const int rc = check_bwd_desc(d, false, "sg_attn_bwd_dq_f16");
    if (rc != SG_OK) return rc;
    const AttnBwdParams p = bwd_params(d);
    hipStream_t st = (hipStream_t)stream;
    if (d->D == 40) launch_bwd<40, 4, SG_BWD_RING, false>(p, st);
    else if (d->D == 80) launch_bwd<80, 4, 2, false>(p, st);
    else launch_bwd<160, 4, 1, false>(p, st);
    SG_CHECK_LAUNCH("sg_attn_bwd_dq_f16");
    return SG_OK;
}

extern "C" int sg_attn_bwd_dkv_f16(const sg_attn_bwd_desc* d, sg_stream_t stream) {
    const int rc = check_bwd_desc(d, true, "sg_attn_bwd_dkv_f16");
    if (rc != SG_OK) return rc;
    const AttnBwdParams p = bwd_params(d);
    hipStream_t st = (hipStream_t)stream;
    if (d->D == 40) launch_bwd<40, 4, SG_BWD_RING, true>(p, st);
    else if (d->D == 80) launch_bwd<80, 4, SG_BWD_RING, true>(p, st);
    else {                                            // 83 KB per stage: a single stage; dK and dV in two passes (registers)
        launch_bwd<160, 4, 1, true, 1>(p, st);
        launch_bwd<160, 4, 1, true, 2>(p, st);
    }
    SG_CHECK_LAUNCH("sg_attn_bwd_dkv_f16");
    return SG_OK;
}

extern "C" int sg_attn_bwd_prep_f32(const sg_half* o, int64_t ldo, int64_t bso, const sg_half* dout, int64_t lddo, int64_t bsdo,
                                    const float* lse2, float* ld2, int32_t B, int32_t H, int32_t Nq, int32_t D, sg_stream_t stream) {
    SG_REQUIRE(o && dout && lse2 && ld2, "sg_attn_bwd_prep: null pointer");
    SG_REQUIRE(B > 0 && H > 0 && Nq > 0 && D > 0 && D % 8 == 0, "sg_attn_bwd_prep: bad shape");
    SG_REQUIRE(ldo % 8 == 0 && lddo % 8 == 0 && bso % 8 == 0 && bsdo % 8 == 0 && sg_aligned16(o) && sg_aligned16(dout) &&
                   (reinterpret_cast<uintptr_t>(ld2) & 7u) == 0, "sg_attn_bwd_prep: strides / alignment");
    const long total = (long)B * H * Nq;
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((int)min((long)4096, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f16*>(o), (long)ldo, (long)bso, reinterpret_cast<const f16*>(dout), (long)lddo,
                       (long)bsdo, lse2, ld2, B, H, Nq, D);
    SG_CHECK_LAUNCH("sg_attn_bwd_prep_f32");
    return SG_OK;
}
