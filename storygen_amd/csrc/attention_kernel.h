// Attention forward kernel template and its launcher (design notes: attention.hip).
#pragma once
#include "common.h"
#include <stdlib.h>

namespace sgattn {


constexpr int KVBLK = 64;            // keys per tile
constexpr float RESCALE_THR = 6.0f;  // log2 units

struct AttnParams {
    const f16* q; long ldq, bsq;
    const f16* k; long ldk, bsk;
    const f16* vt; long ldvt, bsvt;
    f16* o; long ldo, bso;
    int B, H, Nq, Nk, kv_batches, nqb;
    // optional SHORT K / V rows (sg_attn_desc.k2 ...): K/V rows [0, kv2) live in (k2, vt2) with Nk2 keys each, K/V row j >= kv2 is
    // row j - kv2 of (k, vt) with Nk keys (same ldk / ldvt)
    const f16* k2; long bsk2; const f16* vt2; long bsvt2; int Nk2, kv2;
    float scale_log2;   // scale * log2(e)
    float* lse2;        // LSE instantiations only: [B, H, Nq] log2-domain log-sum-exp rows (max + log2(sum)) for the backward
};

__device__ __forceinline__ void glds16(const f16* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int D>
__device__ __forceinline__ int kswz(int row) {
    return D == 40 ? 0 : (D == 80 ? ((row >> 3) & 1) : ((row >> 2) & 3));
}

// D = 40 fast path (round 4): the padded part of the head dimension carries the softmax bookkeeping, so that the MFMAs do the
// work of 66 of the ~160 VALU instructions of a 64-key tile:
//   * contraction slots d = 40, 41 of S^T = K Q^T: the K fragment holds the constants (1, 1) there and the Q^T fragment
//     (-m_hi, -m_lo), the running row maximum split into two fp16 values — the MFMA delivers s - m directly (Q is pre-multiplied by
//     scale * log2 e once per workgroup), so the exponent needs no FMA;
//   * rows d >= 40 of the O^T tile read a row of ONES instead of a duplicate of row 39: O^T row 40 accumulates the row sum of the
//     fp16 probabilities — exactly the values that multiply V — and is rescaled with the accumulators.  No row-sum adds.
// Both constants live in LDS: the (1, 1, 0, ...) K chunks at the start of the workgroup's LDS (one per 32-key block: the lanes
// hi = 1 of k-step 2 read them instead of the first chunk of the next key row), the ones row behind every V^T tile image.
constexpr int F40_KPAD = 2688;       // bytes in front of the ring: 16-byte constants at 0 and 32 * 80 (the two 32-key blocks)
constexpr int F40_ONES = 128;        // bytes behind every tile image: one V^T row of ones

// LDS of one workgroup: the S-stage ring of (K tile | VT tile) images
template <int D, int S, int SUB>
constexpr int attn_smem_bytes() {
    // (+ 16: the general path of a head dim that is not a multiple of 16 reads one chunk past its last K row; D = 80 / 160 do not)
    return D == 40 ? F40_KPAD + S * SUB * (KVBLK * D * 2 + D * 128 + F40_ONES) : S * SUB * (KVBLK * D * 2 + D * 128) + (D % 16 == 0 ? 0 : 16);
}

// One workgroup's work: `block` of `nblocks` (the launch's own numbering — a paired launch runs two problems in one grid).
#ifdef SG_ATTN_RT_STAGE
constexpr bool UNROLL_STAGES = false;     // A/B build (tools/ab_lib.py): the ring stage as a run-time variable, as in rounds 1-4
#else
constexpr bool UNROLL_STAGES = true;
#endif

// KSPLIT (round 5; D = 160, the 16x16 / 8x8 levels): the NW waves of a workgroup share ONE block of 32 queries and split its KEYS — a
// group of NW tiles is loaded into the single ring stage by all waves, wave w computes tile w of it, and the partial (max, sum, O^T)
// of the waves are merged through LDS at the end.  At Nq <= 256 the plain decomposition leaves 24 - 48 workgroups walking 12-tile
// chains at ~5 k cycles per tile (one wave per SIMD: nothing hides the LDS and DMA latencies of a 40 KB tile); splitting the keys
// across the waves shortens that chain NW-fold for the price of an unpipelined ring (S = 1: load, barrier, compute, barrier).
template <int D, int NW, int S, int SUB, bool PRIO, bool LSE, bool LEAN = false, bool GENERAL = false, bool KSPLIT = false>
__device__ __forceinline__ void attn_fwd_body(const AttnParams& p, char* smem, const int block, const int nblocks) {
    constexpr bool F40 = D == 40 && !GENERAL;   // softmax bookkeeping in the padded head dimension (see F40_KPAD)
    constexpr int DC = D / 8;                   // 16-byte chunks per K row
    constexpr int NDK = (D + 15) / 16;          // MFMA k-steps of S^T (contraction padded to 16)
    constexpr int DT = (D + 31) / 32;           // 32-row tiles of O^T
    constexpr int KROW = D * 2;                 // K LDS row stride in bytes
    constexpr int K_BYTES = KVBLK * KROW;       // = D KiB / 8
    constexpr int V_BYTES = D * 128;
    constexpr int K_SEG = K_BYTES / 1024, V_SEG = V_BYTES / 1024, NSEG = K_SEG + V_SEG;   // 1 KiB = one wave DMA
    constexpr int TSTAGE = K_BYTES + V_BYTES + (F40 ? F40_ONES : 0);   // LDS image of one 64-key tile
    constexpr int STAGE = SUB * TSTAGE;         // a ring stage holds SUB consecutive tiles: one barrier per SUB tiles
    constexpr int RING0 = F40 ? F40_KPAD : 0;   // byte offset of the ring
    constexpr int MAXL = (NSEG + NW - 1) / NW;  // DMA instructions per tile of the busiest wave
    constexpr int REM = NSEG % NW;              // waves < REM issue MAXL, the others MAXL - 1 (REM == 0: all MAXL)
    static_assert(KSPLIT ? (S == 1 && SUB == NW && !LSE && !(D == 40 && !GENERAL)) : (S == 2 || S == 3), "2 or 3 stages; key split: one stage of NW tiles");
    static_assert(KSPLIT || SUB == 1 || S == 2, "multi-tile stages use the 2-stage ring (plain vmcnt(0) waits)");
    static_assert(!KSPLIT || NW * (8 + DT * 64) * 64 <= S * SUB * (KVBLK * D * 2 + D * 128), "the merge images of the waves fit the dead ring");
    static_assert((S - 1) * MAXL < 64, "vmcnt is a 6-bit counter");
    static_assert(RING0 + S * STAGE + (F40 || D % 16 == 0 ? 0 : 16) <= attn_smem_bytes<D, S, SUB>(), "LDS size");
    static_assert(!F40 || (32 * KROW + 16 <= F40_KPAD && F40_KPAD % 128 == 0), "K pad constants of both 32-key blocks");

    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // consecutive logical work items run on the same XCD and share its (private, 4 MiB) L2: order them head-major, so that
    // an XCD works on ONE head across all batches at a time — with H = 8 heads, XCD k owns head k.  Its K/V working set
    // is then one head's keys for the distinct K/V batches (2 x 1.9 MB for the main pass's 12 288-key context, where
    // batches 1 and 2 share a row) instead of three different (batch, head) streams that do not fit.
    const int work = xcd_remap(block, nblocks);
    const int bh = work / p.nqb, qb = work - bh * p.nqb;
    const int h = bh / p.B, b = bh - h * p.B;
    const int kvb = b < p.kv_batches ? b : b - (p.B - p.kv_batches);
    const bool short_row = kvb < p.kv2;         // wave-uniform: a K/V row with its own key count (sg_attn_desc.k2)
    const int Nk = short_row ? p.Nk2 : p.Nk;
    const int q0 = KSPLIT ? qb * 32 : (qb * NW + wave) * 32;
    const f16* Q = p.q + (long)b * p.bsq + (long)h * D;
    const f16* K = (short_row ? p.k2 + (long)kvb * p.bsk2 : p.k + (long)(kvb - p.kv2) * p.bsk) + (long)h * D;
    const f16* VT = (short_row ? p.vt2 + (long)kvb * p.bsvt2 : p.vt + (long)(kvb - p.kv2) * p.bsvt) + (long)h * D * p.ldvt;
    const int nkp8 = (Nk + 7) & ~7;             // VT rows hold finite data up to here (host contract)

    if constexpr (F40) {
        // the constants of the fast path: (1, 1, 0, 0, 0, 0, 0, 0) for the K side of contraction slots 40..47 of both 32-key blocks,
        // a row of ones behind every V^T tile image.  Written once; the first ring barrier publishes them (LDS-DMA never touches them).
        if (t < 8) {
            const unsigned v = (t & 3) == 0 ? 0x3C003C00u : 0u;  // halves (1.0, 1.0), then zeros
            *reinterpret_cast<unsigned*>(smem + ((t >> 2) ? 32 * KROW : 0) + 4 * (t & 3)) = v;
        }
        for (int i = t; i < S * SUB * (F40_ONES / 4); i += 64 * NW) {
            const int img = i / (F40_ONES / 4), w = i - img * (F40_ONES / 4);
            *reinterpret_cast<unsigned*>(smem + RING0 + img * TSTAGE + K_BYTES + V_BYTES + 4 * w) = 0x3C003C00u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ---- per-lane DMA source coordinates of this wave's segments (segment g = i*NW + wave).  off = 32-bit element
    // offset from the (wave-uniform) tile origin, so the loads can use an SGPR base + VGPR offset.
    // (row, col) of segment g for this lane — K: key within tile, chunk*8 | VT: d*ldvt, chunk*8.  Only the fast path's
    // combined offset is kept in registers (MAXL = 20 for D = 160); the tail path recomputes the pair.
    auto coords = [&](int g, int& row, int& col) __attribute__((always_inline)) {
        row = col = 0;
        if (g < K_SEG) {
            const int s = g * 64 + lane;        // linear 16-byte slot of the K image
            const int kl = s / DC, cs = s - kl * DC;
            row = kl;
            col = (cs ^ kswz<D>(kl)) * 8;
        } else if (g < NSEG) {
            const int s = (g - K_SEG) * 64 + lane;
            const int d = s >> 3, cs = s & 7;
            row = d * (int)p.ldvt;
            col = (cs ^ ((d >> 1) & 7)) * 8;
        }
    };
    unsigned off[MAXL];
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
        const int g = i * NW + wave;
        int row, col;
        coords(g, row, col);
        off[i] = g < K_SEG ? (unsigned)(row * (int)p.ldk + col) : (unsigned)(row + col);
    }
    const int ntiles = (Nk + KVBLK - 1) / KVBLK;
    // (always_inline: at D = 160 the bodies are large enough for the inliner to leave real calls, which pins `off`
    // and the Q fragments in scratch memory — 6x slower)
    auto issue_tile = [&](int tile, char* base) __attribute__((always_inline)) {
        const int key0 = tile * KVBLK;
        if (key0 + KVBLK <= Nk) {               // full tile: no clamping, uniform base + per-lane offset
            const f16* Kt = K + (long)key0 * p.ldk;
            const f16* Vt = VT + key0;
#pragma unroll
            for (int i = 0; i < MAXL; ++i) {
                const int g = i * NW + wave;    // wave-uniform: the source is a scalar select, not a branch per segment
                const f16* src = (g < K_SEG ? Kt : Vt) + off[i];
                if ((i + 1) * NW <= NSEG || g < NSEG) glds16(src, base + g * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MAXL; ++i) {
                const int g = i * NW + wave;
                int row, col;
                coords(g, row, col);
                if (g < K_SEG) {
                    const int key = min(key0 + row, Nk - 1);                    // tail rows: duplicates (finite)
                    glds16(K + (long)key * p.ldk + col, base + g * 1024);
                } else if (g < NSEG) {
                    const int kc = min(key0 + col, nkp8 - 8);                   // tail chunks: duplicates (finite)
                    glds16(VT + row + kc, base + g * 1024);
                }
            }
        }
    };
    auto issue = [&](int group, int stage) __attribute__((always_inline)) {     // the SUB tiles of ring group `group`
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub)
            if (group * SUB + sub < ntiles) issue_tile(group * SUB + sub, smem + RING0 + stage * STAGE + sub * TSTAGE);
    };

    // ---- Q^T fragments: lane = (query l31, d-chunk 2s+hi); rows beyond Nq are clamped (never stored).  F40: pre-multiplied by
    // scale * log2 e (one more fp16 rounding of q, 2^-11 relative: what the operand already carries), so that S^T is in the log2 domain
    f16x8 qf[NDK];
    {
        const int qi = min(q0 + l31, p.Nq - 1);
#pragma unroll
        for (int s = 0; s < NDK; ++s) {
            const int d0 = s * 16 + hi * 8;
            H8 x; x.u = make_uint4(0, 0, 0, 0);
            if (d0 < D) x.u = ldg16(Q + (long)qi * p.ldq + d0);
            if constexpr (F40) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x.h[j] = (f16)((float)x.h[j] * p.scale_log2);
            }
            qf[s] = x.v;
        }
    }
    // settle the Q loads here: inside the tile loop the only vector-memory traffic must be the LDS-DMA ring, whose
    // counted waits a compiler-inserted vmcnt(0) for these registers would otherwise drain every iteration
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched

    f32x16 oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    // general path: m_run = running max of the scaled (log2-domain) scores of query l31, l_run = this lane's share of the row sum.
    // F40: m_run = the value the Q^T pad slots currently subtract (exactly fp16 hi + fp16 lo; 0 until the first tile has set it);
    // the row sum lives in O^T row 40.
    float m_run = F40 ? 0.f : -INFINITY;
    float l_run = 0.f;

    const int ngroups = (ntiles + SUB - 1) / SUB;
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < ngroups) issue(s, s);

    // LDS read coordinates
    const int prow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);          // pi(l31): swap bits 2 and 3
    // The ring stage is a compile-time constant inside the loop body (S copies of it per trip): the LDS addresses of the fragment reads
    // are then per-lane bases + immediates and the DMA destinations constants, where a run-time stage cost 16 v_add_u32 and a dozen scalar
    // instructions per tile (UNROLL_STAGES = false: the round-1..4 loop with a run-time stage, for A/B builds).  D = 40 only — measured
    // -3.0 / -3.6 % per launch there (309.7 vs 319.1 us at B3 Nq 4096 Nk 12288, 152.6 vs 158.3 at B4 Nk 4096), nothing at D = 80 / 160,
    // whose tile bodies are 2 - 4x larger (profiles/r04l_attention_unrolled_stages.txt).
    constexpr bool UNR = UNROLL_STAGES && D == 40;
    int rt_stage = 0;
    for (int group0 = 0; group0 < ngroups; group0 += (UNR ? S : 1)) {
#pragma unroll
    for (int ustage = 0; ustage < (UNR ? S : 1); ++ustage) {
        const int group = group0 + ustage;
        const int stage = UNR ? ustage : rt_stage;
        if (group < ngroups) {
        if constexpr (KSPLIT) {
            // one stage: every wave has left the previous group's tiles -> load this group's NW tiles -> everybody's share has landed
            if (group > 0) __builtin_amdgcn_s_barrier();
            issue(group, 0);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        } else {
        // wait for this wave's share of the group (S = 3: one younger tile may stay in flight), publish, refill the ring
        if (S == 3 && group + 1 < ngroups) {
            if (REM == 0 || wave < REM) wait_vmcnt<MAXL>();
            else wait_vmcnt<(MAXL > 1 ? MAXL - 1 : 0)>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (group + S - 1 < ngroups) {
            int st = stage + S - 1;
            if (st >= S) st -= S;
            issue(group + S - 1, st);
        }
        }
#pragma unroll
      for (int subi = 0; subi < (KSPLIT ? 1 : SUB); ++subi) {
        const int sub = KSPLIT ? wave : subi;      // key split: this wave's tile of the group
        const int tile = group * SUB + sub;
        if (tile < ntiles) {      // (no `break`: it keeps the per-lane arrays from being promoted to registers)
        const char* sK = smem + RING0 + stage * STAGE + sub * TSTAGE;
        const char* sV = sK + K_BYTES;

        // ---- S^T = K Q^T for the two 32-key blocks.  D <= 80: all K fragments are requested first, so the LDS latency is
        // paid once (40 VGPRs at D = 80); D = 160 would need 80 VGPRs for that and spill, so it reads them per k-step.
        f32x16 s[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        constexpr bool KBATCH = NDK <= 5;
        if constexpr (KBATCH) {
            f16x8 kf[2][NDK];
            if constexpr (F40) {
                // lanes hi = 1 of k-step 2 hold contraction slots 40..47: the constant chunk instead of the next row's first bytes
                const char* k01 = sK + prow * KROW + hi * 16;
                const char* k2 = hi ? smem : sK + prow * KROW + 64;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    kf[kb][0] = *reinterpret_cast<const f16x8*>(k01 + kb * 32 * KROW);
                    kf[kb][1] = *reinterpret_cast<const f16x8*>(k01 + kb * 32 * KROW + 32);
                    kf[kb][2] = *reinterpret_cast<const f16x8*>(k2 + kb * 32 * KROW);
                }
            } else {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const int row = kb * 32 + prow;
                    const char* krow = sK + row * KROW;
                    const int sw = kswz<D>(row);
#pragma unroll
                    for (int st = 0; st < NDK; ++st)
                        kf[kb][st] = *reinterpret_cast<const f16x8*>(krow + (((st * 2 + hi) ^ sw) << 4));
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the MFMAs (the scheduler would re-serialise them)
            // PRIO: raise this wave's issue priority over its SIMD neighbours (other workgroups, in their softmax VALU
            // phase) for the duration of a pure-MFMA cluster — guide T5
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int st = 0; st < NDK; ++st)
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][st], qf[st], st == 0 ? zero16 : s[kb], 0, 0, 0);   // C = inline 0
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        } else {
            // D = 160: the K fragments of all 10 k-steps would need 80 VGPRs, so they are read per k-step — ONE k-step ahead of the MFMAs
            // that consume them (two register sets; the sched_barrier keeps the scheduler from sinking every read to just before its
            // use, which exposed the full LDS latency on each of the 10 k-steps: round 4, as in ff_fused.hip)
            const char* krow0 = sK + prow * KROW;
            const char* krow1 = sK + (32 + prow) * KROW;
            const int sw0 = kswz<D>(prow), sw1 = kswz<D>(32 + prow);
            f16x8 k0 = *reinterpret_cast<const f16x8*>(krow0 + ((hi ^ sw0) << 4));
            f16x8 k1 = *reinterpret_cast<const f16x8*>(krow1 + ((hi ^ sw1) << 4));
#pragma unroll
            for (int st = 0; st < NDK; ++st) {
                f16x8 k0n = k0, k1n = k1;
                if (st + 1 < NDK) {
                    k0n = *reinterpret_cast<const f16x8*>(krow0 + ((((st + 1) * 2 + hi) ^ sw0) << 4));
                    k1n = *reinterpret_cast<const f16x8*>(krow1 + ((((st + 1) * 2 + hi) ^ sw1) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
                s[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[st], st == 0 ? zero16 : s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[st], st == 0 ? zero16 : s[1], 0, 0, 0);
                k0 = k0n; k1 = k1n;
            }
        }
        // ---- V^T fragments of this tile are independent of the softmax: request them now so that their LDS latency
        // hides behind the softmax VALU work (D <= 80: 32 / 48 VGPRs; D = 160 reads them per k-step instead)
        constexpr bool VPRE = DT <= 3 && !LEAN;   // LEAN: V^T fragments per k-step (24 fewer VGPRs: 4 waves per SIMD at D = 40)
        constexpr int DLAST = F40 ? D : D - 1;   // F40: rows >= D read the row of ones behind the image (row sum), else duplicates
        f16x8 vf[VPRE ? 4 : 1][DT];
        if constexpr (VPRE) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    const int d = min(i * 32 + l31, DLAST);      // rows >= D: never stored
                    vf[ks][i] = *reinterpret_cast<const f16x8*>(sV + d * 128 + (((ks * 2 + hi) ^ ((d >> 1) & 7)) << 4));
                }
        }
        // ---- online softmax: mask the key tail, row max (raw scores), deferred rescale, exponentiate
        if ((tile + 1) * KVBLK > Nk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = tile * KVBLK + kb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= Nk) s[kb][r] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        if constexpr (F40) {
            // s holds (log2-domain score) - m_run already.  The first tile always sets the maximum (m_run = 0 is a placeholder there).
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));          // finite: every tile holds >= 1 valid key
            if (tile == 0 || __builtin_amdgcn_ballot_w64(mx > RESCALE_THR) != 0) {
                float m_new = m_run + (tile == 0 ? mx : fmaxf(mx, 0.f));
                m_new = fminf(fmaxf(m_new, -60000.f), 60000.f);
                const f16 mh = (f16)m_new;
                const f16 ml = (f16)(m_new - (float)mh);
                m_new = (float)mh + (float)ml;                // what the pad slots will subtract, exactly
                const float delta = m_new - m_run;
                if (tile != 0) {                              // (tile 0: the accumulators are zero, and exp2(-delta) may overflow)
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int i = 0; i < DT; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
                m_run = m_new;
                if (hi) { qf[NDK - 1][0] = -mh; qf[NDK - 1][1] = -ml; }
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]);
        } else {
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * p.scale_log2;   // finite: every tile holds >= 1 valid key
            if (__builtin_amdgcn_ballot_w64(mx - m_run > RESCALE_THR) != 0) {
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);            // 0 on the first tile (m_run = -inf)
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < DT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            }
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[kb][r], p.scale_log2, -m_run));
                    s[kb][r] = e;
                    psum += e;
                }
            l_run += psum;
        }

        // ---- O^T += VT P^T : 4 k-steps of 16 keys; B fragment = this lane's own P registers
        if constexpr (!VPRE) {      // (D = 160, LEAN) the first k-step's V^T fragments; every later set is requested one k-step ahead
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const int d = min(i * 32 + l31, DLAST);
                vf[0][i] = *reinterpret_cast<const f16x8*>(sV + d * 128 + ((hi ^ ((d >> 1) & 7)) << 4));
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (f16)s[ks >> 1][(ks & 1) * 8 + j];
            f16x8 vn[VPRE ? 1 : DT];
            if constexpr (!VPRE) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < DT; ++i) {
                        const int d = min(i * 32 + l31, DLAST);
                        vn[i] = *reinterpret_cast<const f16x8*>(sV + d * 128 + ((((ks + 1) * 2 + hi) ^ ((d >> 1) & 7)) << 4));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < DT; ++i)
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[VPRE ? ks : 0][i], pf, oacc[i], 0, 0, 0);
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            if constexpr (!VPRE) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < DT; ++i) vf[0][i] = vn[i];
                }
            }
        }
        }
      }   // sub-tiles of the group
        }
        if (!UNR && ++rt_stage == S) rt_stage = 0;
    }
    }

    // ---- normalise and store O[b, q, h*D + d]  (lane holds d = 32i + (r&3) + 8(r>>2) + 4hi for its query).  F40: the row sum is
    // O^T row 40 (hi = 0) / 44 (hi = 1) — both rows of ones saw every key — i.e. register 4 of the second tile in every lane.
    float l_tot;
    if constexpr (F40) l_tot = oacc[DT - 1][4];
    else l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if constexpr (KSPLIT) {
        // merge the waves' partial results for the same 32 queries: wave w > 0 leaves (m, l, O^T) in the dead ring, wave 0 adds them with
        // the usual two-maximum rescaling.  A wave that had no tile holds m = -inf, l = 0, O = 0 and contributes exp2(-inf) = 0.
        constexpr int IMG = (8 + DT * 64) * 64;                       // bytes per wave image: m, l per lane + DT x 16 accumulators per lane
        __builtin_amdgcn_s_barrier();                                 // the last group's tiles have been read by everybody
        if (wave > 0) {
            float* img = reinterpret_cast<float*>(smem + (wave - 1) * IMG);
            img[lane] = m_run;
            img[64 + lane] = l_tot;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) img[128 + (i * 16 + r) * 64 + lane] = oacc[i][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave > 0) return;
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float* img = reinterpret_cast<const float*>(smem + (w - 1) * IMG);
            const float m_o = img[lane], l_o = img[64 + lane];
            const float m_new = fmaxf(m_run, m_o);
            const float fa = __builtin_amdgcn_exp2f(m_run - m_new), fb = __builtin_amdgcn_exp2f(m_o - m_new);
            l_tot = l_tot * fa + l_o * fb;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] = oacc[i][r] * fa + img[128 + (i * 16 + r) * 64 + lane] * fb;
        }
    }
    const float inv = 1.0f / l_tot;
    const int qi = q0 + l31;
    if constexpr (LSE) {    // training forward: P = exp2(s * scale_log2 - lse2) is what the backward kernels recompute
        if (hi == 0 && qi < p.Nq) p.lse2[((long)b * p.H + h) * p.Nq + qi] = m_run + __log2f(l_tot);
    }
    if (qi < p.Nq) {
        f16* O = p.o + (long)b * p.bso + (long)qi * p.ldo + (long)h * D;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = i * 32 + 8 * g + 4 * hi;
                if (d0 < D) {
                    f16x4 w = {(f16)(oacc[i][4 * g + 0] * inv), (f16)(oacc[i][4 * g + 1] * inv),
                               (f16)(oacc[i][4 * g + 2] * inv), (f16)(oacc[i][4 * g + 3] * inv)};
                    *reinterpret_cast<f16x4*>(O + d0) = w;
                }
            }
    }
}

template <int D, int NW, int S, int SUB = 1, bool PRIO = false, bool LSE = false, bool LEAN = false, bool GENERAL = false>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[attn_smem_bytes<D, S, SUB>()];
    attn_fwd_body<D, NW, S, SUB, PRIO, LSE, LEAN, GENERAL>(p, smem, (int)blockIdx.x, (int)gridDim.x);
}

// key-split instantiation (KSPLIT above): one workgroup = NW waves on ONE 32-query block of one (batch, head)
template <int D, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_ksplit_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[attn_smem_bytes<D, 1, NW>()];
    attn_fwd_body<D, NW, 1, NW, false, false, false, false, true>(p, smem, (int)blockIdx.x, (int)gridDim.x);
}

template <int D, int NW>
void launch_attn_ksplit(const AttnParams& p0, hipStream_t st) {
    AttnParams p = p0;
    p.nqb = sg_cdiv(p.Nq, 32);
    hipLaunchKernelGGL((attn_fwd_ksplit_kernel<D, NW>), dim3(p.nqb * p.H * p.B), dim3(64 * NW), 0, st, p);
}

// Two attentions of one transformer block in one grid (text: attention.py:271-276, image: :285-290 — same queries' shape, two
// K/V sources, two outputs): workgroups [0, na) run problem a (the long one: launched first), the rest problem b, which fills the
// CUs that a's tail leaves idle.
template <int D, int NW, int S>
__global__ __launch_bounds__(64 * NW) void attn_fwd_pair_kernel(const AttnParams a, const AttnParams b, const int na) {
    __shared__ __attribute__((aligned(16))) char smem[attn_smem_bytes<D, S, 1>()];
    if ((int)blockIdx.x < na) attn_fwd_body<D, NW, S, 1, false, false>(a, smem, (int)blockIdx.x, na);
    else attn_fwd_body<D, NW, S, 1, false, false>(b, smem, (int)blockIdx.x - na, (int)gridDim.x - na);
}

template <int D, int NW, int S, int SUB = 1, bool PRIO = false, bool LSE = false, bool LEAN = false, bool GENERAL = false>
void launch_attn(const AttnParams& p0, hipStream_t st) {
    AttnParams p = p0;
    p.nqb = sg_cdiv(p.Nq, 32 * NW);
    hipLaunchKernelGGL((attn_fwd_kernel<D, NW, S, SUB, PRIO, LSE, LEAN, GENERAL>), dim3(p.nqb * p.H * p.B), dim3(64 * NW), 0, st, p);
}

template <int D, int NW, int S>
void launch_attn_pair(const AttnParams& a0, const AttnParams& b0, hipStream_t st) {
    AttnParams a = a0, b = b0;
    a.nqb = sg_cdiv(a.Nq, 32 * NW); b.nqb = sg_cdiv(b.Nq, 32 * NW);
    const int na = a.nqb * a.H * a.B, nb = b.nqb * b.H * b.B;
    hipLaunchKernelGGL((attn_fwd_pair_kernel<D, NW, S>), dim3(na + nb), dim3(64 * NW), 0, st, a, b, na);
}


}  // namespace sgattn
