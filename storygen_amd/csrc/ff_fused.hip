// Fused GEGLU feed-forward of a BasicTransformerBlock for gfx950 (round 4; SURVEY 8(f) rank 2: cross-layer fusion):
//   y = Linear2( a * gelu_erf(g) ) + b2 + x,   [a | g] = Linear1( LayerNorm(x) ) + b1            (model/attention.py:298-300,381-393,342)
// in ONE launch — the [M, 4C] GEGLU intermediate (21 MB written and read back per block at the 64x64 level) never exists, and
// neither do the LayerNorm output, its statistics table or the fp16 copy of x.
//
// Decomposition.  A wave owns 32 tokens END TO END: it loads its rows of x (fp32), normalises them in registers (exact two-pass
// LayerNorm; gamma is folded into W1, beta and b1 into the vector d1) and keeps the normalised rows as the B-operand fragments of the
// first GEMM for the whole kernel (C/16 fragments = 80 VGPRs at C = 320).  The hidden dimension is walked in chunks of 32 units:
//   GEMM1  [32 values | 32 gates] x 32 tokens  = W1 chunk (64 rows x C)  . xhat^T          2 C/16 MFMAs (v_mfma_f32_32x32x16_f16)
//   GEGLU  g = v * gelu(gate)  in the accumulator registers -> fp16      (d1 = W1 beta + b1 enters GEMM1 as one more k-step)
//   GEMM2  out[C x 32 tokens] += W2[:, chunk] (C rows x 32) . g^T                          2 C/32 MFMAs
// The value / gate rows of W1 are fed to the MFMA in a bit-permuted order (rows 4..7 <-> 8..11 of every 32-row tile: pi of
// repack._pi32, the same trick as the attention kernel's S^T -> P^T hand-over) so that a lane's GEGLU results ARE the B-operand fragment
// of GEMM2: g never leaves its lane, let alone the register file.  The output tile (C x 32 tokens, fp32) stays in the wave's
// accumulators (160 registers at C = 320) until the epilogue adds b2 and the residual x.  ~350 registers per lane: one wave per SIMD,
// four waves (128 tokens) per workgroup, which share nothing but the weight stream.
//
// Weight stream.  All workgroups walk the same 2.4 MB of weights, chunk by chunk, through LDS: storygen_amd/repack.ff_fused_pack stores
// them byte-for-byte as the LDS images (row-swizzled for conflict-free ds_read_b128, rows pre-permuted), so filling a ring slot is a
// linear LDS-DMA copy (global_load_lds, 1 KiB per wave instruction, no VGPR staging).  W1 parts (C/64 slab images + the chunk's 64 d1
// terms as one more 16-deep k-step, 42 KiB) and W2 parts (C x 64 B, 20 KiB) have separate 2-slot rings, because the loop is software-pipelined by one
// chunk: iteration i runs GEMM1 of chunk i — interleaved, k-step by k-step, with the GEGLU arithmetic of chunk i-1, whose VALU
// instructions then issue in the shadow of the MFMAs — and then GEMM2 of chunk i-1.  One workgroup barrier per iteration publishes
// the parts that have landed and retires the slots about to be refilled (2-slot rings: the refill is issued right after it).
//
// gelu uses erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, three orders of magnitude below the fp16 rounding of g that follows);
// the two-GEMM path's epilogue uses erff.  Both are "exact" GELU in the sense of torch's approximate="none".
#include "common.h"
#include <type_traits>

namespace {

constexpr int FF_NW = 4;                       // waves per workgroup (one per SIMD)

struct FfParams {
    const float* x; long ldx;
    const char* wpack;
    const f16* b2;
    f16* y; long ldy;
    int M; float eps;
    int hsplit;                    // 1: blockIdx.y = 0 / 1 takes the first / second half of the hidden chunks (sg_ff_desc.hidden_split)
    unsigned long long* prof;      // experiments library (sg_debug_ff_anatomy): 8 cycle counters per wave
};

__device__ __forceinline__ void ff_glds16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// GELU: 2 gelu(x) = x + |x| erf(|x| / sqrt 2) with erf by Abramowitz-Stegun 7.1.26, erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 +
// p z) (geglu_stage in the kernel; the factor 1/2 is folded into W2 by repack.ff_fused_pack — exact in fp16): 13 VALU instructions per
// element, two of them transcendental.

// VAR bit 0 (SPREAD): the refill of the rings is issued one DMA instruction per k-step of GEMM1 instead of as a burst behind the barrier;
// VAR bit 1 (DEEP): W1 fragment reads run two k-steps ahead of their MFMAs instead of one.  PROF (experiments library): s_memtime stamps
// around the phases of every iteration, summed per wave (sg_debug_ff_anatomy): [0] iterations [1] vmcnt wait [2] barrier [3] DMA issue
// (burst form) + d1 -> accumulators [4] GEMM1 (+ GEGLU of the previous chunk) [5] GEMM2 [6] prologue (entry -> loop) [7] total
template <int KS, int VAR, bool PROF, bool HS = false>      // KS = C / 64; HS = hidden split (two workgroups per 128 tokens)
__global__ __launch_bounds__(64 * FF_NW) void ff_fused_kernel(const FfParams p) {
    constexpr bool SPREAD = (VAR & 1) != 0, DEEP = (VAR & 2) != 0;
    constexpr int C = 64 * KS, NS = 4 * KS /* k-steps of GEMM1 */, NCT = C / 32 /* output-column tiles */, NCH = 4 * C / 32 /* chunks */;
    constexpr int W1_IMG = KS * 8192, W1_PART = W1_IMG + 64 * 32, W2_PART = C * 64;      // (+ the d1 k-step image: 64 rows x 32 B)
    constexpr int W1_SEG = W1_PART / 1024, W2_SEG = W2_PART / 1024;
    constexpr int N1 = (W1_SEG + FF_NW - 1) / FF_NW, N2 = (W2_SEG + FF_NW - 1) / FF_NW;      // DMA instructions per wave and iteration (at most)
    constexpr int CHUNK = W1_PART + W2_PART;
    static_assert(W2_PART % 1024 == 0 && 2 * CHUNK <= 160 * 1024, "two slots of each ring");
    static_assert(NS >= 16 && NS - NCT >= 0 && N1 + N2 <= NS, "GEGLU, the W2 prefetch and the spread refill ride on GEMM1's k-steps");
    __shared__ __attribute__((aligned(16))) char smem[2 * CHUNK];      // [W1 slot 0 | W1 slot 1 | W2 slot 0 | W2 slot 1]
    char* const w1_ring = smem;
    char* const w2_ring = smem + 2 * W1_PART;

    unsigned long long pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_t = 0, pf_entry = 0;
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if constexpr (PROF) {
            const unsigned long long now = __builtin_readcyclecounter();
            pf_acc[slot] += now - pf_t;
            pf_t = now;
        }
    };
    if constexpr (PROF) pf_entry = pf_t = __builtin_readcyclecounter();

    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m0 = blockIdx.x * (32 * FF_NW) + wave * 32;
    // Hidden split (round 5): at M <= 16 k (the main pass: 96 workgroups for 256 CUs) the launch is as long as ONE wave's walk over the 40
    // hidden chunks.  With hsplit a second workgroup (blockIdx.y = 1) takes chunks [NCH / 2, NCH) of the same tokens and writes its partial
    // sum — no bias, no residual — into columns [C, 2C) of y; the consumer (proj_out) contracts [y_a | y_b] with [W | W].
    // (a separate instantiation: without HS the chunk range is the compile-time [0, NCH) of round 4, instruction for instruction)
    const int half = HS ? (int)blockIdx.y : 0;
    const int c0 = HS ? half * (NCH / 2) : 0, c_end = HS ? c0 + NCH / 2 : NCH;

    // One iteration's refill: W1 part of chunk c1 and W2 part of chunk c2 (each only if it exists).  Linear copies; wave w moves the
    // 1 KiB segments w, w + 4, ... of either part: a compile-time list of DMA instructions per wave (j = 0 .. N1 - 1: W1 segments,
    // then N2 W2 segments; the last of either list only for the waves below the remainder), the only vector term of an address
    // being the lane's 16-byte slot.
    const unsigned lane16 = (unsigned)lane * 16u;
    const int wofs = wave * 1024;
    auto issue_one = [&](int c1, int c2, int j) __attribute__((always_inline)) {      // j is a compile-time constant at every call site
        if (j < N1) {
            const int g = j * FF_NW;                           // + wave
            if (c1 < c_end && (g + FF_NW <= W1_SEG || g + wave < W1_SEG))
                ff_glds16(p.wpack + (size_t)c1 * CHUNK + wofs + g * 1024 + lane16, w1_ring + (c1 & 1) * W1_PART + wofs + g * 1024);
        } else if (j < N1 + N2) {
            const int g = (j - N1) * FF_NW;
            if (c2 >= c0 && c2 < c_end && (g + FF_NW <= W2_SEG || g + wave < W2_SEG))
                ff_glds16(p.wpack + (size_t)c2 * CHUNK + W1_PART + wofs + g * 1024 + lane16, w2_ring + (c2 & 1) * W2_PART + wofs + g * 1024);
        }
    };
    auto issue = [&](int c1, int c2) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < N1 + N2; ++j) issue_one(c1, c2, j);
    };
    issue(c0, -1);

    // ---- this wave's 32 token rows: load (fp32), LayerNorm in registers, keep as fp16 B-operand fragments.  Lane (token l31, hi) holds
    // columns 16 s + 8 hi .. + 7 of its row for s = 0 .. NS - 1 (rows beyond M: a clamped duplicate, never stored).
    f16x8 xf[NS];
    {
        const float* xr = p.x + (long)min(m0 + l31, p.M - 1) * p.ldx + 8 * hi;
        float v[NS][8];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(xr + 16 * s), b = *reinterpret_cast<const float4*>(xr + 16 * s + 4);
            v[s][0] = a.x; v[s][1] = a.y; v[s][2] = a.z; v[s][3] = a.w; v[s][4] = b.x; v[s][5] = b.y; v[s][6] = b.z; v[s][7] = b.w;
        }
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[s][j];
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / C);
        float m2 = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[s][j] - mean; m2 = fmaf(d, d, m2); }
        m2 += __shfl_xor(m2, 32, 64);
        const float rstd = rsqrtf(m2 * (1.0f / C) + p.eps);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[s][j] = (f16)((v[s][j] - mean) * rstd);
    }
    // settle the plain loads: inside the loop the only vector-memory traffic must be the LDS-DMA ring
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    f32x16 out[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ct][r] = 0.f;
    f32x16 accv[2], accg[2];

    // per-lane LDS offsets (the swizzles of repack.ff_fused_pack): W1 image row r, chunk ks*2+hi -> r*128 + ((chunk ^ ((r>>1)&7)) << 4)
    int w1v[4], w1g[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        w1v[ks] = l31 * 128 + (((ks * 2 + hi) ^ ((l31 >> 1) & 7)) << 4);
        w1g[ks] = (32 + l31) * 128 + (((ks * 2 + hi) ^ (((32 + l31) >> 1) & 7)) << 4);
    }
    // W2 image row n = ct*32 + l31, chunk ks*2+hi -> n*64 + ((chunk ^ ((n>>2)&3)) << 4); (ct*32 >> 2) & 3 == 0, so the swizzle is ct-free
    int w2o[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) w2o[ks] = l31 * 64 + (((ks * 2 + hi) ^ ((l31 >> 2) & 3)) << 4);
    // the extra k-step that brings in the chunk's d1 terms: image rows hold (hi, lo, 0, ...) fp16 pairs, the activation side is the
    // constant fragment (1, 1, 0, ...) (lanes hi = 1 hold k = 8..15: zeros on both sides)
    f16x8 xone = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    if (!hi) { xone[0] = (f16)1.f; xone[1] = (f16)1.f; }
    const int dvo = W1_IMG + l31 * 32 + hi * 16, dgo = W1_IMG + (32 + l31) * 32 + hi * 16;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    stamp(6);

    // iteration i, accumulator parity P = i & 1 (compile-time: the two sets are named registers)
    auto body = [&](int i, auto ptag) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's share of W1(i) and W2(i-1) has landed
        stamp(1);
        __builtin_amdgcn_s_barrier();                            // ... everybody's has; the slots of W1(i-1) / W2(i-2) are free
        stamp(2);
        if constexpr (!SPREAD) issue(i + 1, i);
        const bool g1 = i < c_end, g2 = i > c0;                  // wave-uniform
        const char* w1s = w1_ring + (i & 1) * W1_PART;
        const char* w2s = w2_ring + ((i - 1) & 1) * W2_PART;
        // GEGLU of the previous chunk, four elements at a time in four stages of ~13-20 instructions (four independent dependency
        // chains each: one element alone is a ~100-cycle chain of dependent VALU operations, which an in-order wave cannot hide behind
        // two MFMAs).  Group q = registers 4 q .. 4 q + 3 runs its stages on k-steps 4 q .. 4 q + 3 of GEMM1.
        float gl[16];
        float gx[4], gax[4], gt[4], gsq[4], gp[4], ge[4];
        auto geglu_stage = [&](int q, int stage) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (stage == 0) {
                    gx[e] = accg[1 - P][4 * q + e];
                    gax[e] = fabsf(gx[e]);
                    gt[e] = __builtin_amdgcn_rcpf(fmaf(gax[e], 0.3275911f * 0.70710678118654752440f, 1.0f));
                    const float zs = gax[e] * 0.84932180028801904272f;      // |x| sqrt(log2(e) / 2): exp(-x^2 / 2) = exp2(-zs^2)
                    gsq[e] = zs * zs;
                } else if (stage == 1) {
                    ge[e] = __builtin_amdgcn_exp2f(-gsq[e]);
                    gp[e] = fmaf(1.061405429f, gt[e], -1.453152027f);
                    gp[e] = fmaf(gp[e], gt[e], 1.421413741f);
                } else if (stage == 2) {
                    gp[e] = fmaf(gp[e], gt[e], -0.284496736f);
                    gp[e] = fmaf(gp[e], gt[e], 0.254829592f);
                    gp[e] *= gt[e];
                } else {
                    const float erf_abs = fmaf(-gp[e], ge[e], 1.0f);        // erf(|x| / sqrt 2), Abramowitz-Stegun 7.1.26
                    gl[4 * q + e] = accv[1 - P][4 * q + e] * fmaf(gax[e], erf_abs, gx[e]);     // a * 2 gelu(x); W2 is packed halved
                }
            }
        };
        f16x8 a2[NCT];          // W2 fragments of GEMM2's k-step 0: requested under GEMM1's last k-steps
        if (g1) {
            // k-step "d": the chunk's d1 terms (W1 beta + b1) enter through one more MFMA per tile — no accumulator writes
            {
                const f16x8 dv = *reinterpret_cast<const f16x8*>(w1s + dvo), dg = *reinterpret_cast<const f16x8*>(w1s + dgo);
                accv[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dv, xone, zero16, 0, 0, 0);
                accg[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dg, xone, zero16, 0, 0, 0);
            }
            stamp(3);
            // fragment reads run AHEAD k-steps ahead of the MFMAs that consume them (a small ring of named registers)
            constexpr int AHEAD = DEEP ? 2 : 1;
            f16x8 fv[AHEAD + 1], fg[AHEAD + 1];
#pragma unroll
            for (int s = 0; s < AHEAD; ++s) {
                fv[s] = *reinterpret_cast<const f16x8*>(w1s + (s >> 2) * 8192 + w1v[s & 3]);
                fg[s] = *reinterpret_cast<const f16x8*>(w1s + (s >> 2) * 8192 + w1g[s & 3]);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + AHEAD < NS) {
                    const char* slab = w1s + ((s + AHEAD) >> 2) * 8192;
                    fv[(s + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const f16x8*>(slab + w1v[(s + AHEAD) & 3]);
                    fg[(s + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const f16x8*>(slab + w1g[(s + AHEAD) & 3]);
                }
                if (g2 && s >= NS - NCT) a2[s - (NS - NCT)] = *reinterpret_cast<const f16x8*>(w2s + (s - (NS - NCT)) * 2048 + w2o[0]);
                // keep the reads AHEAD of the MFMAs: left alone, the scheduler sinks every fragment read to just before its use (one
                // register set, read -> wait -> MFMA: the full LDS latency exposed on each of the 20 k-steps; measured 160 cycles per k-step)
                __builtin_amdgcn_sched_barrier(0);
                accv[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fv[s % (AHEAD + 1)], xf[s], accv[P], 0, 0, 0);
                accg[P] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fg[s % (AHEAD + 1)], xf[s], accg[P], 0, 0, 0);
                if constexpr (SPREAD) issue_one(i + 1, i, s);           // one refill instruction per k-step (s >= N1 + N2: none)
                if (g2 && s < 16) geglu_stage(s >> 2, s & 3);           // in the shadow of the two MFMAs
            }
        } else if (g2) {
            stamp(3);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) a2[ct] = *reinterpret_cast<const f16x8*>(w2s + ct * 2048 + w2o[0]);
#pragma unroll
            for (int s = 0; s < 16; ++s) geglu_stage(s >> 2, s & 3);
        }
        stamp(4);
        if (g2) {
            f16x8 pf0, pf1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { pf0[j] = (f16)gl[j]; pf1[j] = (f16)gl[8 + j]; }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                out[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[ct], pf0, out[ct], 0, 0, 0);
                a2[ct] = *reinterpret_cast<const f16x8*>(w2s + ct * 2048 + w2o[1]);        // k-step 1's fragment into the freed registers
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) out[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[ct], pf1, out[ct], 0, 0, 0);
        }
        stamp(5);
        if constexpr (PROF) pf_acc[0] += 1;
    };
    static_assert(NCH % 4 == 0, "the loop is unrolled by two, over all chunks or over either half of them");
    for (int i = c0; i < c_end; i += 2) {
        body(i, std::integral_constant<int, 0>{});
        body(i + 1, std::integral_constant<int, 1>{});
    }
    body(c_end, std::integral_constant<int, 0>{});

    // ---- epilogue: y = out + b2 + x (fp32 residual), fp16.  Lane (token l31, hi) register r of tile ct = column ct*32 + 8 (r>>2) + 4 hi + (r&3)
    // (hidden split: the second half writes its partial sum alone, into columns [C, 2C))
    const int m = m0 + l31;
    if (m < p.M) {
        const float* xr = p.x + (long)m * p.ldx + 4 * hi;
        f16* yr = p.y + (long)m * p.ldy + half * C + 4 * hi;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            float4 res[4];
            f16x4 bb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                res[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                bb[q] = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                if (half == 0) {
                    res[q] = *reinterpret_cast<const float4*>(xr + ct * 32 + 8 * q);
                    bb[q] = *reinterpret_cast<const f16x4*>(p.b2 + ct * 32 + 8 * q + 4 * hi);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f16x4 o = {(f16)(out[ct][4 * q + 0] + (float)bb[q][0] + res[q].x), (f16)(out[ct][4 * q + 1] + (float)bb[q][1] + res[q].y),
                                 (f16)(out[ct][4 * q + 2] + (float)bb[q][2] + res[q].z), (f16)(out[ct][4 * q + 3] + (float)bb[q][3] + res[q].w)};
                *reinterpret_cast<f16x4*>(yr + ct * 32 + 8 * q) = o;
            }
        }
    }
    if constexpr (PROF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && p.prof) {
            unsigned long long* dst = p.prof + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * FF_NW + wave) * 8;
#pragma unroll
            for (int k = 0; k < 7; ++k) dst[k] = pf_acc[k];
            dst[7] = __builtin_readcyclecounter() - pf_entry;
        }
    }
}

}  // namespace

extern "C" size_t sg_ff_fused_pack_bytes(int32_t C) {
    if (C != 320) return 0;
    return (size_t)(4 * C / 32) * ((size_t)(C / 64) * 8192 + 64 * 32 + (size_t)C * 64);
}

namespace {
thread_local unsigned long long* g_ff_prof = nullptr;
}

extern "C" int sg_ff_geglu_fused_f16(const sg_ff_desc* d, sg_stream_t stream) {
    SG_REQUIRE(d != nullptr, "sg_ff_geglu_fused_f16: null descriptor");
    SG_REQUIRE(d->x && d->wpack && d->b2 && d->y, "sg_ff_geglu_fused_f16: null pointer");
    if (d->C != 320) return sg_set_error(SG_EUNSUP, "sg_ff_geglu_fused_f16: built for C = 320 (got %d)", d->C);
    SG_REQUIRE(d->M > 0 && d->eps > 0.f, "sg_ff_geglu_fused_f16: bad M / eps");
    SG_REQUIRE(d->hidden_split == 0 || d->hidden_split == 2, "sg_ff_geglu_fused_f16: hidden_split must be 0 or 2");
    const int ycols = d->hidden_split ? 2 * d->C : d->C;
    SG_REQUIRE(d->ldx % 4 == 0 && d->ldx >= d->C && d->ldy % 4 == 0 && d->ldy >= ycols, "sg_ff_geglu_fused_f16: bad ldx / ldy");
    SG_REQUIRE(sg_aligned16(d->x) && sg_aligned16(d->wpack) && sg_aligned16(d->b2) && (reinterpret_cast<uintptr_t>(d->y) & 7u) == 0,
               "sg_ff_geglu_fused_f16: alignment");
    SG_REQUIRE(d->wpack_bytes >= sg_ff_fused_pack_bytes(d->C), "sg_ff_geglu_fused_f16: wpack holds %zu bytes, need %zu", d->wpack_bytes,
               sg_ff_fused_pack_bytes(d->C));
    FfParams p{};
    p.x = d->x; p.ldx = d->ldx; p.wpack = reinterpret_cast<const char*>(d->wpack); p.b2 = reinterpret_cast<const f16*>(d->b2);
    p.y = reinterpret_cast<f16*>(d->y); p.ldy = d->ldy; p.M = d->M; p.eps = d->eps;
    p.hsplit = d->hidden_split ? 1 : 0;
    p.prof = g_ff_prof;
    const dim3 grid(sg_cdiv(d->M, 32 * FF_NW), p.hsplit ? 2 : 1), block(64 * FF_NW);
    hipStream_t st = (hipStream_t)stream;
    const int var = sg_options().ff_variant & 3;        // development option: refill placement / fragment prefetch depth (default 3: both)
#ifdef SG_BUILD_EXPERIMENTS
    if (p.prof) {
        if (p.hsplit) hipLaunchKernelGGL((ff_fused_kernel<5, 3, true, true>), grid, block, 0, st, p);
        else if (var == 0) hipLaunchKernelGGL((ff_fused_kernel<5, 0, true>), grid, block, 0, st, p);
        else if (var == 1) hipLaunchKernelGGL((ff_fused_kernel<5, 1, true>), grid, block, 0, st, p);
        else if (var == 2) hipLaunchKernelGGL((ff_fused_kernel<5, 2, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((ff_fused_kernel<5, 3, true>), grid, block, 0, st, p);
        SG_CHECK_LAUNCH("sg_ff_geglu_fused_f16 (anatomy)");
        return SG_OK;
    }
#endif
    if (p.hsplit) hipLaunchKernelGGL((ff_fused_kernel<5, 3, false, true>), grid, block, 0, st, p);      // (one schedule: the default one)
    else if (var == 0) hipLaunchKernelGGL((ff_fused_kernel<5, 0, false>), grid, block, 0, st, p);
    else if (var == 1) hipLaunchKernelGGL((ff_fused_kernel<5, 1, false>), grid, block, 0, st, p);
    else if (var == 2) hipLaunchKernelGGL((ff_fused_kernel<5, 2, false>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((ff_fused_kernel<5, 3, false>), grid, block, 0, st, p);
    SG_CHECK_LAUNCH("sg_ff_geglu_fused_f16");
    return SG_OK;
}

// Anatomy of the fused feed-forward (experiments library only): the same launch through the instrumented instantiation; prof receives
// 8 uint64 per wave ([workgroup][wave][8], see ff_fused_kernel).
extern "C" int sg_debug_ff_anatomy(const sg_ff_desc* d, void* prof, size_t prof_bytes, sg_stream_t stream) {
#ifdef SG_BUILD_EXPERIMENTS
    SG_REQUIRE(d && prof && prof_bytes >= (size_t)sg_cdiv(d->M, 32 * FF_NW) * FF_NW * 64 * (d->hidden_split ? 2 : 1), "sg_debug_ff_anatomy: need 64 bytes per wave");
    g_ff_prof = reinterpret_cast<unsigned long long*>(prof);
    const int rc = sg_ff_geglu_fused_f16(d, stream);
    g_ff_prof = nullptr;
    return rc;
#else
    (void)d; (void)prof; (void)prof_bytes; (void)stream;
    return sg_set_error(SG_EINVAL, "sg_debug_ff_anatomy: this library was built without SG_BUILD_EXPERIMENTS");
#endif
}
