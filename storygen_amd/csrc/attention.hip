// Fused (flash-style) attention forward for gfx950: O = softmax(scale * Q K^T) V, fp16 in/out, fp32 softmax.
//
// Serves the three attentions of StoryGen's BasicTransformerBlock (model/attention.py:255-260 self, :271-276 text,
// :285-290 image / Visual-Language Context) — head dim D = C/8 in {40, 80, 160}, Nk = HW, 77 or R*HW.
//
// Operands: Q and K token-major ([B, N, H*D], heads interleaved in the channel dimension exactly as
// head_to_batch_dim expects) and V *transposed*: VT[b][h*D + d][key].  The V projection is a GEMM anyway, so the host
// simply computes it with swapped operands (VT = Wv . X^T) — the attention kernel then needs no in-kernel transpose
// and every LDS tile it consumes is a plain 16-byte-chunk copy of global memory, which is what LDS-DMA wants.
//
// Work decomposition: one workgroup = NW wave64, each wave owns 32 queries of one (batch, head); K / VT are streamed in
// tiles of 64 keys through an S-stage LDS ring filled by LDS-DMA (global_load_lds, 16 B per lane, no VGPR staging),
// counted vmcnt waits and ONE raw s_barrier per tile (same pipeline as the GEMM mainloop).
//
// MFMA formulation (v_mfma_f32_32x32x16_f16), chosen so that softmax never leaves registers:
//   S^T[key, q] = sum_d K[key, d] Q[q, d]      A = K fragment (LDS, ds_read_b128), B = Q^T fragment (registers)
//     The A rows are fed in a permuted order (row i of the MFMA reads key pi(i), pi = swap of bits 2 and 3), so that
//     accumulator register r of lane (q = l & 31, hi = l >> 5) holds key 16 (r >> 3) + 8 hi + (r & 7): the 8 registers
//     of a 16-key step are 8 CONSECUTIVE keys.  Row max / row sum are in-lane reductions plus ONE exchange with l^32.
//   O^T[d, q] = sum_key VT[d, key] P^T[key, q]  A = VT fragment (LDS, one ds_read_b128 of 8 consecutive keys),
//     B = P^T = the lane's own accumulator registers converted to fp16.  P never moves across lanes or through LDS.
//   Head dim 40 is zero-padded to 48 on the contraction side (the Q fragment of chunk 5 is zero; the K fragment reads
//   the first 16 bytes of the next LDS row, finite data) and to 64 on the O^T rows (rows >= D are never stored).
// LDS images (both filled by LDS-DMA, so linear in lane order; the swizzles are applied on the SOURCE address and again
// on the read, guide §5.4 rule 21):
//   K tile  [64 keys][D halves], row stride 2D bytes, chunk c of row k at slot c ^ kswz(k): conflict-free ds_read_b128
//           (kswz = 0 for D=40 whose 80-byte stride already spreads 16 rows over 16 slots, (k>>3)&1 for D=80,
//           (k>>2)&3 for D=160).
//   VT tile [D rows][64 keys], row stride 128 bytes, chunk c of row d at slot c ^ ((d>>1)&7).
// Online softmax runs in the log2 domain with the scale folded into the exponent's FMA, and rescales the accumulators
// only when some row's running max grows by more than 2^6 (wave-uniform branch; P <= 64 stays exact enough in fp16
// and the row sums are fp32).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int KVBLK = 64;            // keys per tile
constexpr float RESCALE_THR = 6.0f;  // log2 units

struct AttnParams {
    const f16* q; long ldq, bsq;
    const f16* k; long ldk, bsk;
    const f16* vt; long ldvt, bsvt;
    f16* o; long ldo, bso;
    int B, H, Nq, Nk, kv_batches, nqb;
    float scale_log2;   // scale * log2(e)
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

template <int D, int NW, int S, int SUB = 1>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const AttnParams p) {
    constexpr int DC = D / 8;                   // 16-byte chunks per K row
    constexpr int NDK = (D + 15) / 16;          // MFMA k-steps of S^T (contraction padded to 16)
    constexpr int DT = (D + 31) / 32;           // 32-row tiles of O^T
    constexpr int KROW = D * 2;                 // K LDS row stride in bytes
    constexpr int K_BYTES = KVBLK * KROW;       // = D KiB / 8
    constexpr int V_BYTES = D * 128;
    constexpr int K_SEG = K_BYTES / 1024, V_SEG = V_BYTES / 1024, NSEG = K_SEG + V_SEG;   // 1 KiB = one wave DMA
    constexpr int TSTAGE = K_BYTES + V_BYTES;   // LDS image of one 64-key tile
    constexpr int STAGE = SUB * TSTAGE;         // a ring stage holds SUB consecutive tiles: one barrier per SUB tiles
    constexpr int MAXL = (NSEG + NW - 1) / NW;  // DMA instructions per tile of the busiest wave
    constexpr int REM = NSEG % NW;              // waves < REM issue MAXL, the others MAXL - 1 (REM == 0: all MAXL)
    static_assert(S == 2 || S == 3, "2 or 3 stages");
    static_assert(SUB == 1 || S == 2, "multi-tile stages use the 2-stage ring (plain vmcnt(0) waits)");
    static_assert((S - 1) * MAXL < 64, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) char smem[S * STAGE + 16];

    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // consecutive logical work items run on the same XCD and share its (private, 4 MiB) L2: order them head-major, so that
    // an XCD works on ONE head across all batches at a time — with H = 8 heads, XCD k owns head k.  Its K/V working set
    // is then one head's keys for the distinct K/V batches (2 x 1.9 MB for the main pass's 12 288-key context, where
    // batches 1 and 2 share a row) instead of three different (batch, head) streams that do not fit.
    const int work = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = work / p.nqb, qb = work - bh * p.nqb;
    const int h = bh / p.B, b = bh - h * p.B;
    const int kvb = b < p.kv_batches ? b : b - (p.B - p.kv_batches);
    const int q0 = (qb * NW + wave) * 32;
    const f16* Q = p.q + (long)b * p.bsq + (long)h * D;
    const f16* K = p.k + (long)kvb * p.bsk + (long)h * D;
    const f16* VT = p.vt + (long)kvb * p.bsvt + (long)h * D * p.ldvt;
    const int nkp8 = (p.Nk + 7) & ~7;           // VT rows hold finite data up to here (host contract)

    // ---- per-lane DMA source coordinates of this wave's segments (segment g = i*NW + wave).  off = 32-bit element
    // offset from the (wave-uniform) tile origin, so the loads can use an SGPR base + VGPR offset.
    int c_row[MAXL], c_col[MAXL];               // K: key within tile, chunk*8 | VT: d*ldvt, chunk*8
    unsigned off[MAXL];
#pragma unroll
    for (int i = 0; i < MAXL; ++i) {
        const int g = i * NW + wave;
        c_row[i] = c_col[i] = 0;
        off[i] = 0;
        if (g < K_SEG) {
            const int s = g * 64 + lane;        // linear 16-byte slot of the K image
            const int kl = s / DC, cs = s - kl * DC;
            c_row[i] = kl;
            c_col[i] = (cs ^ kswz<D>(kl)) * 8;
            off[i] = (unsigned)(kl * (int)p.ldk + c_col[i]);
        } else if (g < NSEG) {
            const int s = (g - K_SEG) * 64 + lane;
            const int d = s >> 3, cs = s & 7;
            c_row[i] = d * (int)p.ldvt;
            c_col[i] = (cs ^ ((d >> 1) & 7)) * 8;
            off[i] = (unsigned)(c_row[i] + c_col[i]);
        }
    }
    const int ntiles = (p.Nk + KVBLK - 1) / KVBLK;
    auto issue_tile = [&](int tile, char* base) {
        const int key0 = tile * KVBLK;
        if (key0 + KVBLK <= p.Nk) {             // full tile: no clamping, uniform base + per-lane offset
            const f16* Kt = K + (long)key0 * p.ldk;
            const f16* Vt = VT + key0;
#pragma unroll
            for (int i = 0; i < MAXL; ++i) {
                const int g = i * NW + wave;    // wave-uniform
                if (g < K_SEG) glds16(Kt + off[i], base + g * 1024);
                else if (g < NSEG) glds16(Vt + off[i], base + g * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MAXL; ++i) {
                const int g = i * NW + wave;
                if (g < K_SEG) {
                    const int key = min(key0 + c_row[i], p.Nk - 1);             // tail rows: duplicates (finite)
                    glds16(K + (long)key * p.ldk + c_col[i], base + g * 1024);
                } else if (g < NSEG) {
                    const int kc = min(key0 + c_col[i], nkp8 - 8);              // tail chunks: duplicates (finite)
                    glds16(VT + c_row[i] + kc, base + g * 1024);
                }
            }
        }
    };
    auto issue = [&](int group, int stage) {     // the SUB tiles of ring group `group`
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub)
            if (group * SUB + sub < ntiles) issue_tile(group * SUB + sub, smem + stage * STAGE + sub * TSTAGE);
    };

    // ---- Q^T fragments: lane = (query l31, d-chunk 2s+hi); rows beyond Nq are clamped (never stored)
    f16x8 qf[NDK];
    {
        const int qi = min(q0 + l31, p.Nq - 1);
#pragma unroll
        for (int s = 0; s < NDK; ++s) {
            const int d0 = s * 16 + hi * 8;
            H8 x; x.u = make_uint4(0, 0, 0, 0);
            if (d0 < D) x.u = ldg16(Q + (long)qi * p.ldq + d0);
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
    float m_run = -INFINITY;   // running max of the scaled (log2-domain) scores of query l31
    float l_run = 0.f;         // this lane's share of the running row sum

    const int ngroups = (ntiles + SUB - 1) / SUB;
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < ngroups) issue(s, s);

    // LDS read coordinates
    const int prow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);          // pi(l31): swap bits 2 and 3
    int stage = 0;
    for (int group = 0; group < ngroups; ++group) {
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
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        const int tile = group * SUB + sub;
        if (tile >= ntiles) break;
        const char* sK = smem + stage * STAGE + sub * TSTAGE;
        const char* sV = sK + K_BYTES;

        // ---- S^T = K Q^T for the two 32-key blocks: all K fragments are requested first, so the LDS latency is paid once
        f32x16 s[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f16x8 kf[2][NDK];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int row = kb * 32 + prow;
            const char* krow = sK + row * KROW;
            const int sw = kswz<D>(row);
#pragma unroll
            for (int st = 0; st < NDK; ++st)
                kf[kb][st] = *reinterpret_cast<const f16x8*>(krow + (((st * 2 + hi) ^ sw) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the MFMAs (the scheduler would re-serialise them)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < NDK; ++st)
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][st], qf[st], st == 0 ? zero16 : s[kb], 0, 0, 0);   // C = inline 0
        // ---- V^T fragments of this tile are independent of the softmax: request them now so that their LDS latency
        // hides behind the softmax VALU work (D <= 80: 32 / 48 VGPRs; D = 160 reads them per k-step instead)
        constexpr bool VPRE = DT <= 3;
        f16x8 vf[VPRE ? 4 : 1][DT];
        if constexpr (VPRE) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    const int d = min(i * 32 + l31, D - 1);      // rows >= D: duplicates, never stored
                    vf[ks][i] = *reinterpret_cast<const f16x8*>(sV + d * 128 + (((ks * 2 + hi) ^ ((d >> 1) & 7)) << 4));
                }
        }
        // ---- online softmax: mask the key tail, row max (raw scores), deferred rescale, exponentiate
        if ((tile + 1) * KVBLK > p.Nk) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = tile * KVBLK + kb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= p.Nk) s[kb][r] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
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

        // ---- O^T += VT P^T : 4 k-steps of 16 keys; B fragment = this lane's own P registers
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (f16)s[ks >> 1][(ks & 1) * 8 + j];
            if constexpr (!VPRE) {
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    const int d = min(i * 32 + l31, D - 1);
                    vf[0][i] = *reinterpret_cast<const f16x8*>(sV + d * 128 + (((ks * 2 + hi) ^ ((d >> 1) & 7)) << 4));
                }
            }
#pragma unroll
            for (int i = 0; i < DT; ++i)
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[VPRE ? ks : 0][i], pf, oacc[i], 0, 0, 0);
        }
      }   // sub-tiles of the group
        if (++stage == S) stage = 0;
    }

    // ---- normalise and store O[b, q, h*D + d]  (lane holds d = 32i + (r&3) + 8(r>>2) + 4hi for its query)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qi = q0 + l31;
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

template <int D, int NW, int S, int SUB = 1>
void launch_attn(const AttnParams& p0, hipStream_t st) {
    AttnParams p = p0;
    p.nqb = sg_cdiv(p.Nq, 32 * NW);
    hipLaunchKernelGGL((attn_fwd_kernel<D, NW, S, SUB>), dim3(p.nqb * p.H * p.B), dim3(64 * NW), 0, st, p);
}

}  // namespace

extern "C" int sg_attn_fwd_f16(const sg_attn_desc* d, sg_stream_t stream) {
    SG_REQUIRE(d != nullptr, "sg_attn_fwd_f16: null descriptor");
    SG_REQUIRE(d->q && d->k && d->vt && d->o, "sg_attn_fwd_f16: null q/k/vt/o");
    SG_REQUIRE(d->B > 0 && d->H > 0 && d->Nq > 0 && d->Nk > 0, "sg_attn_fwd_f16: bad shape");
    if (d->D != 40 && d->D != 80 && d->D != 160)
        return sg_set_error(SG_EUNSUP, "sg_attn_fwd_f16: head dim %d not in {40, 80, 160}", d->D);
    SG_REQUIRE(d->kv_batches >= 0 && d->kv_batches <= d->B, "sg_attn_fwd_f16: kv_batches must be in [0, B]");
    SG_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldvt % 8 == 0 && d->ldo % 4 == 0, "sg_attn_fwd_f16: row strides");
    SG_REQUIRE(d->bsq % 8 == 0 && d->bsk % 8 == 0 && d->bsvt % 8 == 0 && d->bso % 4 == 0, "sg_attn_fwd_f16: batch strides");
    SG_REQUIRE(sg_aligned16(d->q) && sg_aligned16(d->k) && sg_aligned16(d->vt) && sg_aligned16(d->o), "sg_attn_fwd_f16: 16-byte alignment");
    const int64_t hd = (int64_t)d->H * d->D;
    SG_REQUIRE(d->ldq >= hd && d->ldk >= hd && d->ldo >= hd, "sg_attn_fwd_f16: token stride smaller than H*D");
    SG_REQUIRE(d->ldvt >= ((d->Nk + 7) & ~7), "sg_attn_fwd_f16: ldvt must cover Nk rounded up to 8 keys");
    SG_REQUIRE((int64_t)d->D * d->ldvt < (1ll << 31), "sg_attn_fwd_f16: VT head slab too large for 32-bit offsets");
    AttnParams p{};
    p.q = reinterpret_cast<const f16*>(d->q); p.ldq = d->ldq; p.bsq = d->bsq;
    p.k = reinterpret_cast<const f16*>(d->k); p.ldk = d->ldk; p.bsk = d->bsk;
    p.vt = reinterpret_cast<const f16*>(d->vt); p.ldvt = d->ldvt; p.bsvt = d->bsvt;
    p.o = reinterpret_cast<f16*>(d->o); p.ldo = d->ldo; p.bso = d->bso;
    p.B = d->B; p.H = d->H; p.Nq = d->Nq; p.Nk = d->Nk;
    p.kv_batches = d->kv_batches > 0 ? d->kv_batches : d->B;
    p.scale_log2 = d->scale * 1.44269504088896340736f;
    hipStream_t st = (hipStream_t)stream;
    // 4-wave workgroups with a 3-deep ring when that still gives the chip >= 2 workgroups per CU, else 2 waves / 2 stages
    const long wgs4 = (long)sg_cdiv(d->Nq, 128) * d->H * d->B;
    const bool big = wgs4 >= 512;
    static const int sub2 = [] { const char* e = getenv("SG_ATTN_SUB2"); return e ? atoi(e) : 0; }();   // development knob
    if (d->D == 40) {
        if (big && sub2 && d->Nk >= 256) launch_attn<40, 4, 2, 2>(p, st);   // 128 keys per barrier, 2-stage ring
        else if (big) launch_attn<40, 4, 3>(p, st);
        else launch_attn<40, 2, 2>(p, st);
    }
    else if (d->D == 80) { if (big) launch_attn<80, 4, 3>(p, st); else launch_attn<80, 2, 2>(p, st); }
    else launch_attn<160, 2, 2>(p, st);
    SG_CHECK_LAUNCH("sg_attn_fwd_f16");
    return SG_OK;
}
