// Fused (flash-style) attention forward for gfx950: O = softmax(scale * Q K^T) V, fp16 in/out, fp32 softmax.
//
// Serves the three attentions of StoryGen's BasicTransformerBlock (model/attention.py:255-260 self, :271-276 text,
// :285-290 image / Visual-Language Context) — head dim D = C/8 in {40, 80, 160}, Nk = HW, 77 or R*HW.
//
// Work decomposition: one 256-thread workgroup = 4 wave64 = 128 query rows of one (batch, head); each wave owns 32
// queries.  K/V are streamed in tiles of 64 keys through LDS (shared by the 4 waves).
//
// MFMA formulation (v_mfma_f32_32x32x16_f16), chosen so that softmax never leaves registers:
//   S^T[key, q]  = sum_d K[key, d] Q[q, d]        A-operand = K fragment (LDS, ds_read_b128), B = Q^T (registers)
//     -> lane l holds query q = l & 31 and 16 of the 32 keys of a block: key = (r&3) + 8*(r>>2) + 4*(l>>5).
//        Row max / row sum are in-lane reductions plus ONE exchange with lane l^32.
//   O^T[d, q]    = sum_key V^T[d, key] P^T[key, q]   A = V^T fragment (LDS, 2x ds_read_b64), B = P^T (registers)
//     -> the MFMA contraction index is permutation-invariant as long as A and B agree, so the k-slot (hi, j) of
//        step ks is *defined* as key 16ks + 4hi + (j&3) + 8(j>>2): exactly the registers the lane already holds
//        after S^T.  P never moves across lanes and never touches LDS.
//   V is transposed on the way into LDS (each thread owns the 8-channel chunks of 4 consecutive keys and emits
//   8-byte stores of V^T[d][4 keys]); its LDS row stride (68 halves) makes the V^T fragment reads conflict-free.
//   K rows are padded to DK+8 halves so the ds_read_b128 of 16 consecutive keys hit 16 distinct 16-B slots.
//   Head dim 40 is zero-padded to 48 on the contraction side (K chunk 5 / Q chunk 5) and to 64 on the O^T rows.
// Pipelining: global loads of tile t+1 are issued into registers before the MFMAs of tile t (two barriers/tile).
#include "common.h"

namespace {

constexpr int QBLK = 128;   // queries per workgroup
constexpr int KVBLK = 64;   // keys per tile
constexpr int VSTR = 68;    // V^T LDS row stride in halves (64 keys + 4 pad)

struct AttnParams {
    const f16* q; long ldq, bsq;
    const f16* k; long ldk, bsk;
    const f16* v; long ldv, bsv;
    f16* o; long ldo, bso;
    int H, Nq, Nk;
    float scale_log2;   // scale * log2(e)
};

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
    constexpr int DK = (D + 15) / 16 * 16;      // contraction length of S^T (padded)
    constexpr int NDK = DK / 16;                // MFMA k-steps for S^T
    constexpr int DT = (D + 31) / 32;           // 32-row tiles of O^T
    constexpr int KSTR = DK + 8;                // K LDS row stride (halves)
    constexpr int DC = D / 8;                   // 16-byte chunks per K/V row
    constexpr int KCH = KVBLK * DC;             // K chunks per tile
    constexpr int K_IT = (KCH + 255) / 256;
    constexpr int VGRP = (KVBLK / 4) * DC;      // V groups (4 keys x one chunk) per tile
    constexpr int V_IT = (VGRP + 255) / 256;
    __shared__ __attribute__((aligned(16))) f16 sK[KVBLK * KSTR];
    __shared__ __attribute__((aligned(16))) f16 sVt[DT * 32 * VSTR];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QBLK + wave * 32;
    const f16* Q = p.q + (long)b * p.bsq + (long)h * D;
    const f16* K = p.k + (long)b * p.bsk + (long)h * D;
    const f16* V = p.v + (long)b * p.bsv + (long)h * D;

    // zero the LDS padding once (never overwritten by the tile stores)
    for (int i = t; i < KVBLK * KSTR; i += 256) sK[i] = (f16)0.f;
    for (int i = t; i < DT * 32 * VSTR; i += 256) sVt[i] = (f16)0.f;

    // Q^T fragments: lane = (query l31, d-chunk 2s+hi)
    f16x8 qf[NDK];
    {
        const int qi = q0 + l31;
#pragma unroll
        for (int s = 0; s < NDK; ++s) {
            const int d0 = s * 16 + hi * 8;
            H8 x; x.u = make_uint4(0, 0, 0, 0);
            if (qi < p.Nq && d0 < D) x.u = ldg16(Q + (long)qi * p.ldq + d0);
            qf[s] = x.v;
        }
    }

    uint4 kreg[K_IT];
    uint4 vreg[V_IT][4];
    auto load_tile = [&](int tile) {
        const int key0 = tile * KVBLK;
#pragma unroll
        for (int i = 0; i < K_IT; ++i) {
            const int ci = t + 256 * i;
            const int key = ci / DC, dc = ci - key * DC;
            kreg[i] = (ci < KCH && key0 + key < p.Nk) ? ldg16(K + (long)(key0 + key) * p.ldk + dc * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < V_IT; ++i) {
            const int gi = t + 256 * i;
            const int kg = gi / DC, dc = gi - kg * DC;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = key0 + kg * 4 + e;
                vreg[i][e] = (gi < VGRP && key < p.Nk) ? ldg16(V + (long)key * p.ldv + dc * 8) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < K_IT; ++i) {
            const int ci = t + 256 * i;
            const int key = ci / DC, dc = ci - key * DC;
            if (ci < KCH) *reinterpret_cast<uint4*>(sK + key * KSTR + dc * 8) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < V_IT; ++i) {
            const int gi = t + 256 * i;
            const int kg = gi / DC, dc = gi - kg * DC;
            if (gi < VGRP) {
                H8 r0, r1, r2, r3;
                r0.u = vreg[i][0]; r1.u = vreg[i][1]; r2.u = vreg[i][2]; r3.u = vreg[i][3];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f16x4 w = {r0.h[j], r1.h[j], r2.h[j], r3.h[j]};
                    *reinterpret_cast<f16x4*>(sVt + (dc * 8 + j) * VSTR + kg * 4) = w;
                }
            }
        }
    };

    f32x16 oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY;   // running max of the scaled (log2-domain) scores of query l31
    float l_run = 0.f;         // this lane's share of the running row sum

    const int ntiles = (p.Nk + KVBLK - 1) / KVBLK;
    load_tile(0);
    for (int tile = 0; tile < ntiles; ++tile) {
        __syncthreads();   // everyone is done reading the previous tile (and the zero fill is complete)
        store_tile();
        __syncthreads();
        if (tile + 1 < ntiles) load_tile(tile + 1);

        // ---- S^T = K Q^T for the two 32-key blocks
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int st = 0; st < NDK; ++st) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(sK + (kb * 32 + l31) * KSTR + (st * 2 + hi) * 8);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[st], s[kb], 0, 0, 0);
            }
        }
        // ---- online softmax (log2 domain): scale, mask the key tail, row max, exponentiate
        const bool tail = (tile + 1) * KVBLK > p.Nk;
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = s[kb][r] * p.scale_log2;
                if (tail) {
                    const int key = tile * KVBLK + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= p.Nk) x = -INFINITY;
                }
                s[kb][r] = x;
                mx = fmaxf(mx, x);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);          // finite: every tile holds >= 1 valid key
        const float alpha = exp2f(m_run - m_new);      // 0 on the first tile
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = exp2f(s[kb][r] - m_new);
                s[kb][r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;

        // ---- O^T += V^T P^T : 4 k-steps of 16 keys; B fragment = this lane's own P registers
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (f16)s[ks >> 1][(ks & 1) * 8 + j];
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const f16* vrow = sVt + (i * 32 + l31) * VSTR + ks * 16 + hi * 4;
                const f16x4 v0 = *reinterpret_cast<const f16x4*>(vrow);
                const f16x4 v1 = *reinterpret_cast<const f16x4*>(vrow + 8);
                const f16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[i], 0, 0, 0);
            }
        }
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

}  // namespace

extern "C" int sg_attn_fwd_f16(const sg_attn_desc* d, sg_stream_t stream) {
    SG_REQUIRE(d != nullptr, "sg_attn_fwd_f16: null descriptor");
    SG_REQUIRE(d->q && d->k && d->v && d->o, "sg_attn_fwd_f16: null q/k/v/o");
    SG_REQUIRE(d->B > 0 && d->H > 0 && d->Nq > 0 && d->Nk > 0, "sg_attn_fwd_f16: bad shape");
    if (d->D != 40 && d->D != 80 && d->D != 160)
        return sg_set_error(SG_EUNSUP, "sg_attn_fwd_f16: head dim %d not in {40, 80, 160}", d->D);
    SG_REQUIRE(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldv % 8 == 0 && d->ldo % 4 == 0, "sg_attn_fwd_f16: token strides");
    SG_REQUIRE(d->bsq % 8 == 0 && d->bsk % 8 == 0 && d->bsv % 8 == 0 && d->bso % 4 == 0, "sg_attn_fwd_f16: batch strides");
    SG_REQUIRE(sg_aligned16(d->q) && sg_aligned16(d->k) && sg_aligned16(d->v) && sg_aligned16(d->o), "sg_attn_fwd_f16: 16-byte alignment");
    SG_REQUIRE(d->ldq >= (int64_t)d->H * d->D && d->ldk >= (int64_t)d->H * d->D && d->ldv >= (int64_t)d->H * d->D &&
                   d->ldo >= (int64_t)d->H * d->D, "sg_attn_fwd_f16: token stride smaller than H*D");
    AttnParams p{};
    p.q = reinterpret_cast<const f16*>(d->q); p.ldq = d->ldq; p.bsq = d->bsq;
    p.k = reinterpret_cast<const f16*>(d->k); p.ldk = d->ldk; p.bsk = d->bsk;
    p.v = reinterpret_cast<const f16*>(d->v); p.ldv = d->ldv; p.bsv = d->bsv;
    p.o = reinterpret_cast<f16*>(d->o); p.ldo = d->ldo; p.bso = d->bso;
    p.H = d->H; p.Nq = d->Nq; p.Nk = d->Nk;
    p.scale_log2 = d->scale * 1.44269504088896340736f;
    dim3 grid((d->Nq + QBLK - 1) / QBLK, d->H, d->B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (d->D == 40) hipLaunchKernelGGL(attn_fwd_kernel<40>, grid, block, 0, st, p);
    else if (d->D == 80) hipLaunchKernelGGL(attn_fwd_kernel<80>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<160>, grid, block, 0, st, p);
    SG_CHECK_LAUNCH("sg_attn_fwd_f16");
    return SG_OK;
}
