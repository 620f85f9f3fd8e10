// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X, CDNA4), fp16 operands / fp32 accumulate.
//
// One mainloop family serves both entry points (sg_gemm_f16, sg_conv3x3_nhwc_f16): C[M,N] = A[M,K] . W[N,K]^T where,
// for the convolution, row m is an output pixel and the K axis enumerates (ky, kx, ci) — the A tile is *gathered*
// from the NHWC input (optional nearest-2x upsample and stride 2 folded into the index).
//
// Structure (CDNA4-first, see /opt/skills/guides/cdna_hip_programming.md §5):
//   * every wave owns a 64x64 output sub-tile = 2x2 accumulators of v_mfma_f32_32x32x16_f16 (one 16-byte fragment
//     per lane per operand per MFMA, 1 KiB of LDS reads per MFMA); a workgroup is a WGM x WGN grid of such waves
//     (256x128, 128x128, 256x64, 128x64, 64x128 or 64x64 tiles), chosen per problem so that the 1024 SIMDs of the chip stay busy.
//   * pipelined kernel (the fast path): K is walked in 64-deep slabs through S=3 LDS stages filled by LDS-DMA
//     (global_load_lds, 16 B per lane, no VGPR staging), counted vmcnt waits and ONE raw s_barrier per slab.
//   * LDS rows are 128 B (64 halves); the 16-byte chunk c of row r lives at slot c ^ ((r>>1)&7): conflict-free for
//     the ds_read_b128 fragment reads (a 16-lane service group touches 16 distinct 16-B slots of the 256-B bank row).
//     LDS-DMA writes lane l of a wave to (wave-uniform base + 16 l), so the swizzle is applied on the SOURCE side
//     (the lane that owns slot s of row r fetches logical chunk s ^ ((r>>1)&7)) and again on the reads.
//   * generic kernel (fallback): register-staged double buffer with zero-fill predicates, for K % 64 != 0 or an
//     unpadded convolution input.
//   * the fp32 tile is staged through LDS for the epilogue so that bias / residual / output accesses are 16-byte,
//     row-contiguous; epilogue math is fp32; outputs and residuals may be fp16 or fp32 (the UNet's residual stream
//     is kept in fp32, MFMA operands in fp16).
//   * block ids are remapped so that consecutive tiles (same A row panel) run on the same XCD / L2.
//   * small-M layers (16x16 / 8x8 latent levels at batch 3) are split along K into fp32 partial tiles; a second kernel
//     reduces them in slice order and applies the epilogue (deterministic, no atomics).  An in-launch reduction by the
//     last-arriving slice (agent-scope release / acquire + ticket counter) was built and measured in round 2: the release
//     fence behind 64 KB of freshly written partials costs more (+8..13 us per launch) than the kernel boundary it removes.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BK = 64;
constexpr int MAX_AUTO_SPLIT = 16;

struct MmaParams {
    const f16* A; long lda;
    const f16* W; long ldw;
    void* C; long ldc;
    f16* C2; long ldc2;            // optional second (fp16) copy of the output
    int M, N, K, KT;
    // conv geometry (CONV only): input [B,H,Wd,Cin] (pre-upsample), output [B,Ho,Wo,N]
    int H, Wd, Ho, Wo, cpt /* Cin/64 */, stride, ups, padded;
    // epilogue
    int mode, flags;
    const f16* bias;
    const float* rowbias; long rowbias_ld; int rows_per_batch;
    const void* res1; long ldr1;
    const void* res2; long ldr2;
    // decomposition
    float* ws; int splits; int kt_per_split; int tiles_m, tiles_n;
    float* stats; int stats_batch_rows;   // optional GroupNorm partial statistics of the output (tile_epilogue), else nullptr
    unsigned long long* prof;   // PROF instantiations only (sg_debug_*_anatomy): per-wave cycle totals of the mainloop phases
    int n_major;   // tile order: 1 = consecutive ids walk M first (tiles sharing a weight panel stay on one XCD / L2)
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void add_res8(const void* res, long ld, bool f32, int gm, int gn, float (&v)[8]) {
    if (f32) {
        const float* r = reinterpret_cast<const float*>(res) + (long)gm * ld + gn;
        const float4 a = *reinterpret_cast<const float4*>(r), b = *reinterpret_cast<const float4*>(r + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    } else {
        H8 r; r.u = ldg16(reinterpret_cast<const f16*>(res) + (long)gm * ld + gn);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)r.h[j];
    }
}

__device__ __forceinline__ void store_out8(const MmaParams& p, int gm, int gn, const float (&v)[8]) {
    const bool f32 = p.flags & SG_F_OUT_F32;
    if (f32) {
        float* o = reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (!f32 || p.C2) {
        H8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.h[j] = (f16)v[j];
        if (!f32) stg16(reinterpret_cast<f16*>(p.C) + (long)gm * p.ldc + gn, o.u);
        if (p.C2) stg16(p.C2 + (long)gm * p.ldc2 + gn, o.u);
    }
}

__device__ __forceinline__ void epi_linear8(const MmaParams& p, int gm, int gn, float (&v)[8]) {
    if (p.bias) {
        H8 b; b.u = ldg16(p.bias + gn);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)b.h[j];
    }
    if (p.rowbias) {
        const float* rb = p.rowbias + (long)(gm / p.rows_per_batch) * p.rowbias_ld + gn;
        const float4 r0 = *reinterpret_cast<const float4*>(rb), r1 = *reinterpret_cast<const float4*>(rb + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
        v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    if (p.res1) add_res8(p.res1, p.ldr1, p.flags & SG_F_RES1_F32, gm, gn, v);
    if (p.res2) add_res8(p.res2, p.ldr2, p.flags & SG_F_RES2_F32, gm, gn, v);
    store_out8(p, gm, gn, v);
}

// val/gate: 8 consecutive interleaved-layout columns starting at global column gv (value) and gv+32 (gate).
__device__ __forceinline__ void epi_geglu8(const MmaParams& p, int gm, int gv, float (&val)[8], float (&gate)[8]) {
    if (p.bias) {
        H8 bv, bg; bv.u = ldg16(p.bias + gv); bg.u = ldg16(p.bias + gv + 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) { val[j] += (float)bv.h[j]; gate[j] += (float)bg.h[j]; }
    }
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = val[j] * gelu_erf_f(gate[j]);
    store_out8(p, gm, (gv >> 6) * 32 + (gv & 31), o);   // interleaved column -> output column
}

// Shared tail of both mainloops: split-K partial store, or LDS-staged fused epilogue with 16-byte accesses.
// Must be entered by all threads after a barrier that ends all LDS reads of the mainloop.  Wave (wm, wn) of the
// WGM x WGN grid holds TM x TN 32x32 accumulators of its (BM/WGM) x (BN/WGN) sub-tile.
// DUAL (mma_pp_kernel): the workgroup is TWO groups of WGM x WGN waves that each hold a partial accumulator of the same tile
// (even / odd K slabs); a band is staged by group 0 and then added to by group 1, and all 2 x 64 WGM WGN threads consume it.
template <int BM, int BN, int WGM, int WGN, bool DUAL = false>
__device__ __forceinline__ void tile_epilogue(const MmaParams& p, char* smem, f32x16 (&acc)[BM / WGM / 32][BN / WGN / 32],
                                              int m0, int n0, int z) {
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32, NWG = WGM * WGN, NT = (DUAL ? 128 : 64) * NWG;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int group = DUAL ? wave / NWG : 0, wv = DUAL ? wave - group * NWG : wave;
    const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hi = lane >> 5;
    // one 32-row band of every wave row -> sC [WGM*32][BN] fp32 (DUAL: group 0 stores, group 1 adds); ends with a barrier
    auto stage_band = [&](int ip, bool first) __attribute__((always_inline)) {
        float* sC = reinterpret_cast<float*>(smem);
        if (!first) __syncthreads();   // previous band fully consumed
        if (group == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = wn * WN + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) sC[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * BN + n] = acc[ip][j][r];
            }
        }
        __syncthreads();
        if constexpr (DUAL) {
            if (group == 1) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = wn * WN + j * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sC[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * BN + n] += acc[ip][j][r];
                }
            }
            __syncthreads();
        }
    };
    if constexpr (DUAL) {
        if (p.splits > 1) {   // merged partial tile -> workspace, 32 bytes per thread and item
            float* wsz = p.ws + (size_t)z * p.M * p.N;
            constexpr int NCHs = BN / 8, BRs = WGM * 32;
#pragma unroll
            for (int ip = 0; ip < TM; ++ip) {
                stage_band(ip, ip == 0);
                const float* sC = reinterpret_cast<const float*>(smem);
                for (int idx = t; idx < BRs * NCHs; idx += NT) {
                    const int lr = idx / NCHs, ch = idx - lr * NCHs;
                    const int gm = m0 + (lr >> 5) * WM + ip * 32 + (lr & 31), gn = n0 + ch * 8;
                    if (gm >= p.M || gn >= p.N) continue;
                    float* dst = wsz + (size_t)gm * p.N + gn;
                    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(sC + lr * BN + ch * 8);
                    *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(sC + lr * BN + ch * 8 + 4);
                }
            }
            return;
        }
    }
    if (p.splits > 1) {   // raw fp32 partial tile; the epilogue happens in splitk_reduce_kernel
        float* wsz = p.ws + (size_t)z * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = n0 + wn * WN + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (gm < p.M && gn < p.N) wsz[(size_t)gm * p.N + gn] = acc[i][j][r];
                }
            }
        return;
    }
    // ---- fused epilogue through LDS, one 32-row band of every wave's sub-tile at a time (TM passes): the staging
    // buffer is [WGM*32][BN] fp32 = BM*BN*4/TM bytes, so the epilogue never needs more LDS than the mainloop ring and
    // small rings (S = 2) leave room for a second workgroup of another kernel on the same CU.
    float* sC = reinterpret_cast<float*>(smem);
    constexpr int BR = WGM * 32;   // rows staged per pass
    if (p.mode == SG_EPI_LINEAR) {
        // every thread owns IT = 4 (row, 8-column chunk) items per band (BR * BN/8 / NT == 4 for every tile shape).  The
        // additive epilogue terms (bias, temb row-bias, residuals) do not depend on the accumulators, so their global
        // loads are issued BEFORE the band is staged: the HBM latency overlaps the LDS round trip instead of following it.
        constexpr int NCH = BN / 8, IT = BR * NCH / NT;
        static_assert(IT * NT == BR * NCH, "items per thread must be integral");
        static_assert(NT % NCH == 0 && 64 % NCH == 0, "a thread's items share one column chunk");
        const bool want_stats = p.stats != nullptr;
        float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ip = 0; ip < TM; ++ip) {
            float add[IT][8];
            int gmv[IT], gnv[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int idx = t + it * NT;
                const int lr = idx / NCH, ch = idx - lr * NCH;
                gmv[it] = m0 + (lr >> 5) * WM + ip * 32 + (lr & 31);
                gnv[it] = n0 + ch * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) add[it][j] = 0.f;
                if (gmv[it] < p.M && gnv[it] < p.N) {
                    if (p.bias) {
                        H8 b; b.u = ldg16(p.bias + gnv[it]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) add[it][j] += (float)b.h[j];
                    }
                    if (p.rowbias) {
                        const float* rb = p.rowbias + (long)(gmv[it] / p.rows_per_batch) * p.rowbias_ld + gnv[it];
                        const float4 r0 = *reinterpret_cast<const float4*>(rb), r1 = *reinterpret_cast<const float4*>(rb + 4);
                        add[it][0] += r0.x; add[it][1] += r0.y; add[it][2] += r0.z; add[it][3] += r0.w;
                        add[it][4] += r1.x; add[it][5] += r1.y; add[it][6] += r1.z; add[it][7] += r1.w;
                    }
                    if (p.res1) add_res8(p.res1, p.ldr1, p.flags & SG_F_RES1_F32, gmv[it], gnv[it], add[it]);
                    if (p.res2) add_res8(p.res2, p.ldr2, p.flags & SG_F_RES2_F32, gmv[it], gnv[it], add[it]);
                }
            }
            stage_band(ip, ip == 0);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                if (gmv[it] >= p.M || gnv[it] >= p.N) continue;
                const int idx = t + it * NT;
                const int lr = idx / NCH, ch = idx - lr * NCH;
                const float4 v0 = *reinterpret_cast<const float4*>(sC + lr * BN + ch * 8);
                const float4 v1 = *reinterpret_cast<const float4*>(sC + lr * BN + ch * 8 + 4);
                float v[8] = {v0.x + add[it][0], v0.y + add[it][1], v0.z + add[it][2], v0.w + add[it][3],
                              v1.x + add[it][4], v1.y + add[it][5], v1.z + add[it][6], v1.w + add[it][7]};
                store_out8(p, gmv[it], gnv[it], v);
                if (want_stats) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { cs[j] += v[j]; cq[j] = fmaf(v[j], v[j], cq[j]); }
                }
            }
        }
        if (want_stats) {
            // GroupNorm statistics as an epilogue: per-(row tile, channel) sums of the FINAL fp32 values (bias / temb / residual
            // included, before the fp16 rounding).  A thread's items all sit in column chunk t % NCH (NT % NCH == 0), so: lanes
            // with equal lane % NCH -> one lane (shuffles), waves -> LDS -> fixed-order sum: deterministic, no atomics.
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int o = 32; o >= NCH; o >>= 1) {
                    cs[j] += __shfl_xor(cs[j], o, 64);
                    cq[j] += __shfl_xor(cq[j], o, 64);
                }
            }
            __syncthreads();                    // the last band's LDS reads are done: sC can be reused
            float* sred = reinterpret_cast<float*>(smem);            // [wave][NCH][16]
            if (lane < NCH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sred[(wave * NCH + lane) * 16 + j] = cs[j];
                    sred[(wave * NCH + lane) * 16 + 8 + j] = cq[j];
                }
            }
            __syncthreads();
            if (t < NCH * 2) {                   // thread (plane = t / NCH, ch = t % NCH): 8 consecutive columns of one plane
                const int plane = t / NCH, ch = t - plane * NCH;
                const int gn = n0 + ch * 8;
                if (gn < p.N) {
                    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int w = 0; w < NT / 64; ++w)
#pragma unroll
                        for (int j = 0; j < 8; ++j) a[j] += sred[(w * NCH + ch) * 16 + plane * 8 + j];
                    float* dst = p.stats + ((size_t)(m0 / BM) * 2 + plane) * p.N + gn;
                    *reinterpret_cast<float4*>(dst) = make_float4(a[0], a[1], a[2], a[3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(a[4], a[5], a[6], a[7]);
                }
            }
        }
        return;
    }
    // GEGLU: bias only (tiny, cached); one band at a time
#pragma unroll
    for (int ip = 0; ip < TM; ++ip) {
        stage_band(ip, ip == 0);
        constexpr int OCH = BN / 16;
        for (int idx = t; idx < BR * OCH; idx += NT) {
            const int lr = idx / OCH, j = idx - lr * OCH;
            const int vcol = (j >> 2) * 64 + (j & 3) * 8;
            const int gm = m0 + (lr >> 5) * WM + ip * 32 + (lr & 31), gv = n0 + vcol;
            if (gm >= p.M || gv >= p.N) continue;
            const float* sp = sC + lr * BN + vcol;
            const float4 a0 = *reinterpret_cast<const float4*>(sp), a1 = *reinterpret_cast<const float4*>(sp + 4);
            const float4 g0 = *reinterpret_cast<const float4*>(sp + 32), g1 = *reinterpret_cast<const float4*>(sp + 36);
            float val[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            float gate[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            epi_geglu8(p, gm, gv, val, gate);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Generic kernel: 256 threads (2x2 waves), register-staged double buffer, zero-fill predicates.
template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void mma_kernel(const MmaParams p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int EPI_BYTES = 2 * 32 * BN * 4;   // one 32-row band per wave-row (tile_epilogue)
    constexpr int SMEM = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    // logical id = tile * splits + slice: the K slices of a tile are consecutive ids (one XCD, see xcd_remap)
    const int lid2 = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n * p.splits);
    const int lid = lid2 / p.splits, z = lid2 - lid * p.splits;
    const int m0 = (p.n_major ? lid % p.tiles_m : lid / p.tiles_n) * BM;
    const int n0 = (p.n_major ? lid / p.tiles_m : lid % p.tiles_n) * BN;
    const int kt0 = z * p.kt_per_split;
    const int kt1 = min(p.KT, kt0 + p.kt_per_split);

    // per-thread staging coordinates: chunk c of rows r0 + 32*i
    const int c = t & 7, r0 = t >> 3;
    const f16* a_ptr[A_IT];
    int a_oy[A_IT], a_ox[A_IT];
    bool a_ok[A_IT];
    const int pad = p.padded ? 1 : 0;
    const int wrow = p.Wd + 2 * pad;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int gm = m0 + r0 + 32 * i;
        a_ok[i] = gm < p.M;
        if constexpr (CONV) {
            const int hw = p.Ho * p.Wo;
            const int b = gm / hw, rem = gm - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_oy[i] = a_ok[i] ? oy * p.stride - 1 : -(1 << 20);
            a_ox[i] = ox * p.stride - 1;
            a_ptr[i] = p.A + (long)b * (p.H + 2 * pad) * wrow * p.lda + c * 8;
        } else {
            a_oy[i] = a_ox[i] = 0;
            a_ptr[i] = p.A + (long)gm * p.lda + c * 8;
        }
    }
    const f16* w_ptr[B_IT];
    bool w_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int gn = n0 + r0 + 32 * i;
        w_ok[i] = gn < p.N;
        w_ptr[i] = p.W + (long)gn * p.ldw + c * 8;
    }
    const int hin = p.H << p.ups, win = p.Wd << p.ups;

    uint4 areg[A_IT], breg[B_IT];
    auto load_regs = [&](int kt) {
        const bool kok = kt * BK + c * 8 < p.K;
        if constexpr (CONV) {
            const int tap = kt / p.cpt, cc = kt - tap * p.cpt;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int iy = a_oy[i] + ky, ix = a_ox[i] + kx;
                const bool ok = (unsigned)iy < (unsigned)hin && (unsigned)ix < (unsigned)win;
                const long pix = (long)((iy >> p.ups) + pad) * wrow + ((ix >> p.ups) + pad);
                areg[i] = ok ? ldg16(a_ptr[i] + pix * p.lda + cc * BK) : make_uint4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                areg[i] = (a_ok[i] && kok) ? ldg16(a_ptr[i] + kt * BK) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            breg[i] = (w_ok[i] && kok) ? ldg16(w_ptr[i] + kt * BK) : make_uint4(0, 0, 0, 0);
    };
    auto store_lds = [&](int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<uint4*>(sA + lds_off(r0 + 32 * i, c)) = areg[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<uint4*>(sB + lds_off(r0 + 32 * i, c)) = breg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt0 < kt1) {
        load_regs(kt0);
        store_lds(0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        const bool more = kt + 1 < kt1;
        if (more) load_regs(kt + 1);
        const char* sA = smem + buf * STAGE;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f16x8*>(sA + lds_off(wm * WM + i * 32 + l31, ks * 2 + hi));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f16x8*>(sB + lds_off(wn * WN + j * 32 + l31, ks * 2 + hi));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_lds(buf ^ 1);
        __syncthreads();
    }
    tile_epilogue<BM, BN, 2, 2>(p, smem, acc, m0, n0, z);
}

// ------------------------------------------------------------------------------------------------------------
// Pipelined kernel: WGM x WGN waves of 64x64, S LDS stages filled by LDS-DMA.  Requires loads that need no
// predicate: rows beyond M / N are clamped to the last valid row (their results are discarded by the epilogue);
// K % 64 == 0; for the convolution the input has a one-pixel zero border ([B, H+2, W+2, C]), so every tap of every
// output pixel reads valid memory and padding costs nothing.
// Slab t+S-1 is issued right after the barrier that (a) publishes slab t (every wave waited for its own DMA with a
// counted vmcnt first) and (b) retires every wave's reads of slab t-1, whose stage it overwrites.
__device__ __forceinline__ void glds16(const f16* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WGM, int WGN, int S, int WTM = 2, int WTN = 2>
constexpr int pipe_smem_bytes() {
    constexpr int STAGE = (32 * WTM * WGM + 32 * WTN * WGN) * 128, EPI = WGM * 32 * (32 * WTN * WGN) * 4;
    return S * STAGE > EPI ? S * STAGE : EPI;
}

// SPREAD: how the LDS-DMA instructions of the slab two ahead are placed among the 16 MFMAs of the current slab.  0: all of them
// between k-step 0 and k-step 1 (the matrix pipe drains while 6-16 DMA instructions are issued: ~400-550 of ~1400-2000 cycles
// per slab, tools/anatomy.py); 1: a third each in front of k-steps 1, 2, 3; 2: one or two behind every MFMA of k-steps 1-3, order
// pinned with sched_barrier.
template <int WGM, int WGN, int S, bool CONV, bool LATE, bool PREF, int WTM = 2, int WTN = 2, bool PROF = false, int SPREAD = 0>
__device__ __forceinline__ void mma_pipe_body(const MmaParams& p, char* smem) {
    // PROF: s_memtime stamps around the phases of every slab, summed per wave (sg_debug_gemm_anatomy / _conv_anatomy):
    // [0] slabs [1] vmcnt wait [2] barrier [3] first fragment reads + k-step 0 [4] k-step 1 up to the DMA issue [5] DMA issue
    // [6] rest of the slab [7] prologue (entry -> loop) [8] epilogue (loop end -> exit)
    unsigned long long pf_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pf_t = 0, pf_entry = 0;
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if constexpr (PROF) {
            const unsigned long long now = __builtin_readcyclecounter();
            pf_acc[slot] += now - pf_t;
            pf_t = now;
        }
    };
    if constexpr (PROF) pf_entry = pf_t = __builtin_readcyclecounter();
    // every wave owns a (32 WTM) x (32 WTN) output sub-tile.  2 x 2 needs 1 KiB of LDS fragment reads per MFMA, which at
    // full MFMA rate is the whole LDS read bandwidth of the CU (8 waves x 32 B/clk); "fat" 4 x 2 waves (128 x 64, 128
    // accumulator registers, one wave per SIMD) need 0.75 KiB per MFMA and half as many waves for the same tile.
    constexpr int NW = WGM * WGN, WM = 32 * WTM, WN = 32 * WTN, BM = WM * WGM, BN = WN * WGN;
    constexpr int A_IT = BM / (8 * NW), B_IT = BN / (8 * NW), LPT = A_IT + B_IT;   // LDS-DMA instructions / lane / slab
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int EPI_BYTES = WGM * 32 * BN * 4;   // one 32-row band per wave-row (tile_epilogue)
    constexpr int SMEM = (S * STAGE > EPI_BYTES) ? S * STAGE : EPI_BYTES;
    constexpr int ISTR = NW * 1024;   // LDS bytes covered by one DMA instruction of the whole workgroup (8 rows / wave)
    static_assert(S >= 2 && S <= 4 && (S - 2) * LPT < 64, "2 to 4 stages; vmcnt is a 6-bit counter");
    static_assert(S * STAGE <= 160 * 1024, "the ring must fit the 160 KB of LDS");
    static_assert(SMEM == pipe_smem_bytes<WGM, WGN, S, WTM, WTN>(), "LDS size of the wrappers");

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, hi = lane >> 5;
    // logical id = tile * splits + slice: the K slices of a tile are consecutive ids (one XCD, see xcd_remap)
    const int lid2 = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n * p.splits);
    const int lid = lid2 / p.splits, z = lid2 - lid * p.splits;
    const int m0 = (p.n_major ? lid % p.tiles_m : lid / p.tiles_n) * BM;
    const int n0 = (p.n_major ? lid / p.tiles_m : lid % p.tiles_n) * BN;
    const int kt0 = z * p.kt_per_split;
    const int nt = min(p.KT, kt0 + p.kt_per_split) - kt0;

    // staging coordinates of this lane for DMA instruction i: tile row srow + 8*NW*i, LDS slot (lane & 7).
    // All per-lane address arithmetic is done ONCE here as 32-bit element offsets; per slab only wave-uniform (scalar)
    // terms change: GEMM  A + kt*64;  conv  A + ((ky*wp + kx)*lda + cc*64)  — or, with nearest-2x upsampling, where the
    // source row (oy-1+ky)>>1 is not affine in ky, one of three precomputed row / column offsets picked by (ky, kx).
    const int srow = wave * 8 + (lane >> 3);
    const int wp = p.Wd + 2;
    unsigned a_off[A_IT];                       // offset of the row (GEMM) / of tap (0, 0) (conv)
    unsigned a_par[A_IT];                       // conv with upsampling: parity bits of (oy - 1, ox - 1)
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = srow + 8 * NW * i;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);          // logical chunk this lane fetches
        const int gm = min(m0 + row, p.M - 1);
        if constexpr (CONV) {
            const int hw = p.Ho * p.Wo;
            const int b = gm / hw, rem = gm - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const long img = (long)b * (p.H + 2) * wp;
            // padded input pixel of tap (ky, kx): row ((oy*stride - 1 + ky) >> ups) + 1, column likewise
            if (!p.ups) {
                a_off[i] = (unsigned)((img + (long)(oy * p.stride) * wp + ox * p.stride) * p.lda + lc * 8);
                a_par[i] = 0;
            } else {
                // source row of tap ky: ((oy - 1 + ky) >> 1) + 1 = ((oy - 1) >> 1) + 1 + ((ky + ((oy - 1) & 1)) >> 1)
                a_off[i] = (unsigned)((img + (long)(((oy - 1) >> 1) + 1) * wp + ((ox - 1) >> 1) + 1) * p.lda + lc * 8);
                a_par[i] = (unsigned)(((oy - 1) & 1) | (((ox - 1) & 1) << 1));
            }
        } else {
            a_off[i] = (unsigned)((long)gm * p.lda + lc * 8);
            a_par[i] = 0;
        }
    }
    unsigned w_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = srow + 8 * NW * i;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        w_off[i] = (unsigned)((long)min(n0 + row, p.N - 1) * p.ldw + lc * 8);
    }

    // part / nparts: only the loads whose index (A loads first, then W loads) is congruent to `part` modulo `nparts` (all: 0, 1)
    auto issue = [&](int kt, int stage, int part = 0, int nparts = 1) __attribute__((always_inline)) {
        char* sA = smem + stage * STAGE + wave * 1024;
        char* sB = sA + A_BYTES;
        if constexpr (CONV) {
            const int tap = kt / p.cpt, cc = kt - tap * p.cpt;
            const int ky = tap / 3, kx = tap - ky * 3;
            if (!p.ups) {
                const f16* At = p.A + ((long)(ky * wp + kx) * p.lda + cc * BK);
#pragma unroll
                for (int i = 0; i < A_IT; ++i)
                    if (i % nparts == part) glds16(At + a_off[i], sA + i * ISTR);
            } else {
                const f16* At = p.A + cc * BK;
                const unsigned rs = (unsigned)(wp * (int)p.lda), cs = (unsigned)p.lda;     // < 2^24 (validated on the host)
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    if (i % nparts != part) continue;
                    const unsigned dy = ((unsigned)ky + (a_par[i] & 1u)) >> 1, dx = ((unsigned)kx + (a_par[i] >> 1)) >> 1;
                    glds16(At + (a_off[i] + __umul24(dy, rs) + __umul24(dx, cs)), sA + i * ISTR);
                }
            }
        } else {
            const f16* At = p.A + kt * BK;
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                if (i % nparts == part) glds16(At + a_off[i], sA + i * ISTR);
        }
        const f16* Wt = p.W + kt * BK;
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if ((A_IT + i) % nparts == part) glds16(Wt + w_off[i], sB + i * ISTR);
    };

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nt) issue(kt0 + s, s);

    int stage = 0;
    stamp(7);
    for (int it = 0; it < nt; ++it) {
        // up to S - 2 younger slabs stay in flight while we wait for slab `it` (fewer at the very end)
        if (S >= 4 && it + 2 < nt) wait_vmcnt<(S >= 4 ? 2 : 0) * LPT>();
        else if (S >= 3 && it + 1 < nt) wait_vmcnt<(S >= 3 ? 1 : 0) * LPT>();
        else wait_vmcnt<0>();
        stamp(1);
        __builtin_amdgcn_s_barrier();
        stamp(2);
        if (!LATE && it + S - 1 < nt) {         // refill right behind the barrier
            int st = stage + S - 1;
            if (st >= S) st -= S;
            issue(kt0 + it + S - 1, st);
        }
        const char* sA = smem + stage * STAGE;
        const char* sB = sA + A_BYTES;
        // fragment reads run one k-step ahead of the MFMAs that consume them (two register sets), so that only the first
        // read of a slab exposes LDS latency; the other three hide behind the previous k-step's four MFMAs
        f16x8 af[2][WTM], bf[2][WTN];
        auto load_frags = [&](int buf, int ks) {
#pragma unroll
            for (int i = 0; i < WTM; ++i)
                af[buf][i] = *reinterpret_cast<const f16x8*>(sA + lds_off(wm * WM + i * 32 + l31, ks * 2 + hi));
#pragma unroll
            for (int j = 0; j < WTN; ++j)
                bf[buf][j] = *reinterpret_cast<const f16x8*>(sB + lds_off(wn * WN + j * 32 + l31, ks * 2 + hi));
        };
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (PREF) { if (ks + 1 < 4) load_frags((ks + 1) & 1, ks + 1); }
            else if (ks > 0) load_frags(ks & 1, ks);
            // the refill of the ring (slab it+S-1 into the stage every wave has just left) is issued behind the first
            // k-step's fragment reads rather than between the barrier and them: its address arithmetic then overlaps
            // matrix work instead of delaying it
            int st = stage + S - 1;
            if (st >= S) st -= S;
            const bool refill = LATE && it + S - 1 < nt;
            if (SPREAD == 0 && ks == 1 && refill) {
                stamp(4);
                issue(kt0 + it + S - 1, st);
                stamp(5);
            }
            if (SPREAD == 1 && ks >= 1 && refill) issue(kt0 + it + S - 1, st, ks - 1, 3);
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                    if (SPREAD == 2 && ks >= 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (refill) issue(kt0 + it + S - 1, st, (ks - 1) * WTM * WTN + i * WTN + j, 3 * WTM * WTN);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            if (ks == 0) stamp(3);
        }
        if (++stage == S) stage = 0;
        stamp(6);
        if constexpr (PROF) pf_acc[0] += 1;
    }
    __syncthreads();   // every wave is done reading the stages before the epilogue reuses LDS
    tile_epilogue<BM, BN, WGM, WGN>(p, smem, acc, m0, n0, z);
    if constexpr (PROF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(8);
        if (lane == 0 && p.prof) {
            unsigned long long* dst = p.prof + ((size_t)blockIdx.x * NW + wave) * 10;
#pragma unroll
            for (int k = 0; k < 9; ++k) dst[k] = pf_acc[k];
            dst[9] = pf_t - pf_entry;
        }
    }
}

template <int WGM, int WGN, bool CONV, int SPREAD>
__global__ __launch_bounds__(64 * WGM * WGN) void mma_pipe_prof_kernel(const MmaParams p) {
    __shared__ __attribute__((aligned(16))) char smem[pipe_smem_bytes<WGM, WGN, 3>()];
    mma_pipe_body<WGM, WGN, 3, CONV, true, true, 2, 2, true, SPREAD>(p, smem);
}

template <int WGM, int WGN, bool CONV, int SPREAD>
__global__ __launch_bounds__(64 * WGM * WGN) void mma_pipe_spread_kernel(const MmaParams p) {
    __shared__ __attribute__((aligned(16))) char smem[pipe_smem_bytes<WGM, WGN, 3>()];
    mma_pipe_body<WGM, WGN, 3, CONV, true, true, 2, 2, false, SPREAD>(p, smem);
}

// ------------------------------------------------------------------------------------------------------------
// Ping-pong variant of the pipelined kernel (round 2; the answer to tools/anatomy.py's finding that the phases of a slab ADD
// because all waves of a workgroup move in lock step).  The workgroup is TWO groups of WGM x WGN waves working on the SAME
// output tile: group 0 accumulates the even K slabs, group 1 the odd ones, each from its own 2-stage LDS ring filled by its own
// waves.  Every interval between two workgroup barriers one group computes a slab (fragment reads + 16 MFMAs per wave) while the
// other issues the LDS-DMA of its slab after next and waits for its next one — each SIMD hosts one wave of either group, so
// DMA issue, vmcnt wait and barrier latency of one group run beside the matrix work of the other.  The two partial accumulators
// are added in the LDS-staged epilogue (tile_epilogue<..., DUAL>).  Same loads, one more fp32 addition per output: results agree
// with mma_pipe_kernel to fp32 summation order.
// Protocol of group g (its slab i is K slab 2 i + g; n_g slabs; stage = i & 1), identical barrier count for both groups:
//   prologue: issue slab 0 and slab 1; group 0 waits for slab 0.
//   interval k = 0 .. nt-1:  s_barrier;
//       k % 2 == g : compute slab i = k / 2                      (landed: waited for in interval k-1, published by this barrier)
//       else       : i' = (k + 1) / 2 = the slab computed next;  issue slab i'+1 (if >= 2: its stage held slab i'-1, computed in
//                    interval k-1, before this barrier), then wait until slab i' has landed (leaving slab i'+1 in flight).
template <int WGM, int WGN, bool CONV>
__global__ __launch_bounds__(128 * WGM * WGN) void mma_pp_kernel(const MmaParams p) {
    constexpr int NW = WGM * WGN, WM = 64, WN = 64, BM = WM * WGM, BN = WN * WGN;
    constexpr int A_IT = BM / (8 * NW), B_IT = BN / (8 * NW), LPT = A_IT + B_IT;   // LDS-DMA instructions / lane / slab (one group)
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int EPI_BYTES = WGM * 32 * BN * 4;
    constexpr int SMEM = (4 * STAGE > EPI_BYTES) ? 4 * STAGE : EPI_BYTES;
    constexpr int ISTR = NW * 1024;
    static_assert(4 * STAGE <= 160 * 1024, "two 2-stage rings must fit the 160 KB of LDS");
    static_assert(LPT < 64, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = wave / NW, wv = wave - g * NW;               // group, wave within the group
    const int wm = wv / WGN, wn = wv % WGN, l31 = lane & 31, hi = lane >> 5;
    const int lid2 = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n * p.splits);
    const int lid = lid2 / p.splits, z = lid2 - lid * p.splits;
    const int m0 = (p.n_major ? lid % p.tiles_m : lid / p.tiles_n) * BM;
    const int n0 = (p.n_major ? lid / p.tiles_m : lid % p.tiles_n) * BN;
    const int kt0 = z * p.kt_per_split;
    const int nt = min(p.KT, kt0 + p.kt_per_split) - kt0;      // K slabs of this block
    const int ng = (nt - g + 1) >> 1;                           // ... of this group: slabs kt0 + 2 i + g
    char* const ring = smem + g * 2 * STAGE;

    const int srow = wv * 8 + (lane >> 3);
    const int wp = p.Wd + 2;
    unsigned a_off[A_IT], a_par[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = srow + 8 * NW * i;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        const int gm = min(m0 + row, p.M - 1);
        if constexpr (CONV) {
            const int hw = p.Ho * p.Wo;
            const int b = gm / hw, rem = gm - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const long img = (long)b * (p.H + 2) * wp;
            if (!p.ups) {
                a_off[i] = (unsigned)((img + (long)(oy * p.stride) * wp + ox * p.stride) * p.lda + lc * 8);
                a_par[i] = 0;
            } else {
                a_off[i] = (unsigned)((img + (long)(((oy - 1) >> 1) + 1) * wp + ((ox - 1) >> 1) + 1) * p.lda + lc * 8);
                a_par[i] = (unsigned)(((oy - 1) & 1) | (((ox - 1) & 1) << 1));
            }
        } else {
            a_off[i] = (unsigned)((long)gm * p.lda + lc * 8);
            a_par[i] = 0;
        }
    }
    unsigned w_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = srow + 8 * NW * i;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        w_off[i] = (unsigned)((long)min(n0 + row, p.N - 1) * p.ldw + lc * 8);
    }
    auto issue = [&](int i) __attribute__((always_inline)) {   // this group's slab i -> stage i & 1
        const int kt = kt0 + 2 * i + g;
        char* sA = ring + (i & 1) * STAGE + wv * 1024;
        char* sB = sA + A_BYTES;
        if constexpr (CONV) {
            const int tap = kt / p.cpt, cc = kt - tap * p.cpt;
            const int ky = tap / 3, kx = tap - ky * 3;
            if (!p.ups) {
                const f16* At = p.A + ((long)(ky * wp + kx) * p.lda + cc * BK);
#pragma unroll
                for (int j = 0; j < A_IT; ++j) glds16(At + a_off[j], sA + j * ISTR);
            } else {
                const f16* At = p.A + cc * BK;
                const unsigned rs = (unsigned)(wp * (int)p.lda), cs = (unsigned)p.lda;
#pragma unroll
                for (int j = 0; j < A_IT; ++j) {
                    const unsigned dy = ((unsigned)ky + (a_par[j] & 1u)) >> 1, dx = ((unsigned)kx + (a_par[j] >> 1)) >> 1;
                    glds16(At + (a_off[j] + __umul24(dy, rs) + __umul24(dx, cs)), sA + j * ISTR);
                }
            }
        } else {
            const f16* At = p.A + kt * BK;
#pragma unroll
            for (int j = 0; j < A_IT; ++j) glds16(At + a_off[j], sA + j * ISTR);
        }
        const f16* Wt = p.W + kt * BK;
#pragma unroll
        for (int j = 0; j < B_IT; ++j) glds16(Wt + w_off[j], sB + j * ISTR);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (ng > 0) issue(0);
    if (ng > 1) issue(1);
    if (g == 0) {
        if (ng > 1) wait_vmcnt<LPT>();
        else wait_vmcnt<0>();
    }
#pragma unroll 1
    for (int k = 0; k < nt; ++k) {
        __builtin_amdgcn_s_barrier();
        if ((k & 1) == g) {
            const int i = k >> 1;
            const char* sA = ring + (i & 1) * STAGE;
            const char* sB = sA + A_BYTES;
            f16x8 af[2][2], bf[2][2];
            auto load_frags = [&](int buf, int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    af[buf][u] = *reinterpret_cast<const f16x8*>(sA + lds_off(wm * WM + u * 32 + l31, ks * 2 + hi));
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    bf[buf][u] = *reinterpret_cast<const f16x8*>(sB + lds_off(wn * WN + u * 32 + l31, ks * 2 + hi));
            };
            load_frags(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) load_frags((ks + 1) & 1, ks + 1);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int v = 0; v < 2; ++v)
                        acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][u], bf[ks & 1][v], acc[u][v], 0, 0, 0);
            }
        } else {
            const int in = (k + 1) >> 1;                       // the slab this group computes in the next interval
            if (in < ng) {
                const bool more = in + 1 < ng;
                if (more && in + 1 >= 2) issue(in + 1);
                if (more) wait_vmcnt<LPT>();
                else wait_vmcnt<0>();
            }
        }
    }
    __syncthreads();   // every wave is done reading its ring before the epilogue reuses LDS
    tile_epilogue<BM, BN, WGM, WGN, true>(p, smem, acc, m0, n0, z);
}

template <int WGM, int WGN, int S, bool CONV, bool LATE, bool PREF, int WTM = 2, int WTN = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void mma_pipe_kernel(const MmaParams p) {
    __shared__ __attribute__((aligned(16))) char smem[pipe_smem_bytes<WGM, WGN, S, WTM, WTN>()];
    mma_pipe_body<WGM, WGN, S, CONV, LATE, PREF, WTM, WTN>(p, smem);
}

// Two independent GEMMs in ONE launch (blockIdx.y selects the problem; blocks beyond a problem's grid exit): the q|k and V^T
// projections of one LayerNorm output, the text / image query projections, the attn3 K and V^T projections of a finished
// context.  Each pair shares its activation operand and is far too small to fill the chip alone, so the pair costs about one
// launch instead of two (and one dependency boundary instead of two).  Static selection (two inlined bodies): a runtime
// index into the kernel arguments would move them to scratch.
struct MmaPair { MmaParams p0, p1; };

template <int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void mma_pipe_pair_kernel(const MmaPair pp) {
    __shared__ __attribute__((aligned(16))) char smem[pipe_smem_bytes<WGM, WGN, 3>()];
    if (blockIdx.y == 0) {
        if ((int)blockIdx.x < pp.p0.tiles_m * pp.p0.tiles_n * pp.p0.splits) mma_pipe_body<WGM, WGN, 3, false, true, true>(pp.p0, smem);
    } else {
        if ((int)blockIdx.x < pp.p1.tiles_m * pp.p1.tiles_n * pp.p1.splits) mma_pipe_body<WGM, WGN, 3, false, true, true>(pp.p1, smem);
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const MmaParams p) {
    const size_t MN = (size_t)p.M * p.N;
    if (p.mode == SG_EPI_LINEAR) {
        const int nch = p.N / 8;
        const long total = (long)p.M * nch;
        for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
            const int gm = (int)(idx / nch), gn = (int)(idx - (long)gm * nch) * 8;
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const float* s = p.ws + (size_t)gm * p.N + gn;
            // the partial tiles of up to four splits are requested together (one memory round trip instead of four) and
            // added in split order, so the sum is bit-identical to the sequential loop
            for (int z0 = 0; z0 < p.splits; z0 += 4) {
                float4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (z0 + u < p.splits) {
                        a[u] = *reinterpret_cast<const float4*>(s + (z0 + u) * MN);
                        b[u] = *reinterpret_cast<const float4*>(s + (z0 + u) * MN + 4);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[0] += a[u].x; v[1] += a[u].y; v[2] += a[u].z; v[3] += a[u].w;
                    v[4] += b[u].x; v[5] += b[u].y; v[6] += b[u].z; v[7] += b[u].w;
                }
            }
            epi_linear8(p, gm, gn, v);
        }
    } else {
        const int och = p.N / 16;
        const long total = (long)p.M * och;
        for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
            const int gm = (int)(idx / och), j = (int)(idx - (long)gm * och);
            const int gv = (j >> 2) * 64 + (j & 3) * 8;
            float val[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gate[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const float* s = p.ws + (size_t)gm * p.N + gv;
            for (int z0 = 0; z0 < p.splits; z0 += 2) {      // two splits (4 x 16 B each) per memory round trip
                float4 q[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) q[u][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (z0 + u < p.splits) {
                        const float* src = s + (z0 + u) * MN;
                        q[u][0] = *reinterpret_cast<const float4*>(src);
                        q[u][1] = *reinterpret_cast<const float4*>(src + 4);
                        q[u][2] = *reinterpret_cast<const float4*>(src + 32);
                        q[u][3] = *reinterpret_cast<const float4*>(src + 36);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    val[0] += q[u][0].x; val[1] += q[u][0].y; val[2] += q[u][0].z; val[3] += q[u][0].w;
                    val[4] += q[u][1].x; val[5] += q[u][1].y; val[6] += q[u][1].z; val[7] += q[u][1].w;
                    gate[0] += q[u][2].x; gate[1] += q[u][2].y; gate[2] += q[u][2].z; gate[3] += q[u][2].w;
                    gate[4] += q[u][3].x; gate[5] += q[u][3].y; gate[6] += q[u][3].z; gate[7] += q[u][3].w;
                }
            }
            epi_geglu8(p, gm, gv, val, gate);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// 3x3 convolution (stride 1, zero-bordered input) with the INPUT PATCH RESIDENT IN LDS.
//
// The implicit-GEMM kernel above gathers the A tile of every (tap, channel chunk) slab from global memory: each input
// pixel of a tile crosses the CU's vector L1 nine times.  Every convolution of the step is bound by exactly that per-CU
// operand feed, so this kernel moves fewer bytes per FLOP instead: a tile is BM consecutive output
// pixels = BM / W whole image rows (W | BM), whose 3x3 footprint is ONE contiguous range of (BM / W + 2) (W + 2)
// padded input pixels.  For each 64-channel chunk that patch is brought into LDS once (LDS-DMA, double-buffered:
// the patch of chunk c + 1 lands while the nine taps of chunk c are consumed) and the MFMA A fragments of tap (ky, kx) are
// read from it at a pixel offset of ky (W + 2) + kx.  Only the weights stream per slab (S = 3 ring, as above).
// A-side L1 traffic drops from 9 BM x 128 B to (BM / W + 2)(W + 2) x 128 B per chunk (x 5.8 at 256 x 64-pixel rows).
// K order = (channel chunk, tap); split-K splits whole chunks.  LDS pixel rows are 128 B with the same XOR swizzle
// (slot = chunk ^ ((pixel >> 1) & 7), applied on the DMA source side and on the reads).
// vmcnt bookkeeping: every iteration issues ONE group after its barrier — the weight slab two iterations ahead plus, at
// taps 0..6 of every chunk but the last, PT pieces (1 KiB each) of the next patch — and iteration `it` needs everything
// up to group it - 2, i.e. it may leave exactly |group it - 1| loads outstanding; group sizes are the same in every wave
// (surplus pieces copy a valid pixel into a scrap KiB), so the counts are compile-time constants per tap position.
template <int WGM, int WGN>
struct PatchGeom {
    static constexpr int NW = WGM * WGN, BM = 64 * WGM, BN = 64 * WGN;
    static constexpr int B_IT = BN / (8 * NW);
    static constexpr int NPIECES_MAX = ((BM / 64 + 2) * 66 + 7) / 8;              // W <= 64: the largest patch, in 8-pixel pieces
    static constexpr int PP = (NPIECES_MAX + NW - 1) / NW;                         // pieces per wave per chunk
    static constexpr int PT = (PP + 6) / 7;                                        // pieces per wave per tap (taps 0..6)
    static constexpr int PATCH_BYTES = NPIECES_MAX * 1024;
    static constexpr int W_STAGE = BN * 128, S = 3;
    static constexpr int SCRAP = 2 * PATCH_BYTES + S * W_STAGE;
    static constexpr int EPI_BYTES = WGM * 32 * BN * 4;
    static constexpr int SMEM = (SCRAP + 1024 > EPI_BYTES) ? SCRAP + 1024 : EPI_BYTES;
    static_assert(SMEM <= 160 * 1024, "patch buffers + weight ring must fit the 160 KB of LDS");
    static_assert(B_IT + PT < 32, "vmcnt is a 6-bit counter");
};

template <int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_patch_kernel(const MmaParams p) {
    using G = PatchGeom<WGM, WGN>;
    constexpr int NW = G::NW, BM = G::BM, BN = G::BN, B_IT = G::B_IT, PT = G::PT, S = G::S;
    constexpr int ISTR = NW * 1024;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM];
    char* const wring = smem + 2 * G::PATCH_BYTES;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, hi = lane >> 5;
    const int lid2 = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n * p.splits);
    const int lid = lid2 / p.splits, z = lid2 - lid * p.splits;
    const int m0 = (p.n_major ? lid % p.tiles_m : lid / p.tiles_n) * BM;
    const int n0 = (p.n_major ? lid / p.tiles_m : lid % p.tiles_n) * BN;
    const int cps = p.kt_per_split / 9;                       // channel chunks per K slice
    const int c0 = z * cps, nch = min(p.cpt, c0 + cps) - c0;  // this block's chunks [c0, c0 + nch)
    const int nt = nch * 9;

    // geometry: tile = rows [y0, y0 + BM / Wd) of image b; patch = padded rows [y0, y0 + BM / Wd + 2), all Wd + 2 columns
    const int Wd = p.Wd, wp = Wd + 2, hw = p.Ho * p.Wo;
    const int b = m0 / hw, y0 = (m0 - b * hw) / Wd;
    const int np = (BM / Wd + 2) * wp, npieces = (np + 7) >> 3;
    const unsigned pix0 = (unsigned)((b * (p.H + 2) + y0) * wp);          // first padded pixel of the patch

    // weight-slab staging (as in mma_pipe_kernel): tile row srow + 8 NW i, LDS slot lane & 7, source-side swizzle
    const int srow = wave * 8 + (lane >> 3);
    unsigned w_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = srow + 8 * NW * i;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        w_off[i] = (unsigned)((long)min(n0 + row, p.N - 1) * p.ldw + lc * 8);
    }
    // A fragments: patch pixel of tap (0, 0) for this lane's row of each 32-row MFMA tile
    int q0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wm * 64 + i * 32 + l31;
        const int r = m / Wd;
        q0[i] = r * wp + (m - r * Wd);
    }

    auto issue_w = [&](int it) {                               // weight slab of iteration `it` (chunk-major K order)
        const int ci = it / 9, wt = it - ci * 9;               // scalar arithmetic (wave-uniform)
        char* sB = wring + (it % S) * G::W_STAGE + wave * 1024;
        const f16* Wt = p.W + (wt * p.cpt + c0 + ci) * BK;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16(Wt + w_off[i], sB + i * ISTR);
    };
    auto issue_piece = [&](int chunk, int j) {                 // piece j of this wave of the patch of chunk `chunk` (relative)
        const int g = j * NW + wave;                           // wave-uniform piece id
        const int pix = min(g * 8 + (lane >> 3), np - 1);
        const int lc = (lane & 7) ^ ((pix >> 1) & 7);
        const f16* src = p.A + ((size_t)(pix0 + pix) * p.lda + (c0 + chunk) * BK + lc * 8);
        char* dst = (g < npieces) ? smem + (chunk & 1) * G::PATCH_BYTES + g * 1024 : smem + G::SCRAP;
        glds16(src, dst);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: whole patch of chunk 0, weight slabs 0 and 1
#pragma unroll 1
    for (int j = 0; j < 7 * PT; ++j) issue_piece(0, j);
    issue_w(0);
    if (nt > 1) issue_w(1);

    // one rolled loop over the slabs (a 9-way unrolled tap loop hoists nine sets of fragment addresses: > 256 VGPRs);
    // (ch, tap, ky, kx, stage) are wave-uniform counters kept in SGPRs
    int ch = 0, tap = 0, ky = 0, kx = 0, stage = 0;
#pragma unroll 1
    for (int it = 0; it < nt; ++it) {
        const bool more = ch + 1 < nch;                        // a next chunk exists: its patch is prefetched during this one
        // leave exactly the previous iteration's group in flight (see header)
        if (it + 1 >= nt) wait_vmcnt<0>();
        else if (tap == 0 || tap == 8 || !more) wait_vmcnt<B_IT>();
        else wait_vmcnt<B_IT + PT>();
        __builtin_amdgcn_s_barrier();
        const char* pbuf = smem + (ch & 1) * G::PATCH_BYTES;
        const char* sB = wring + stage * G::W_STAGE;
        const int toff = ky * wp + kx;
        int abase[2], aswz[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = q0[i] + toff;
            abase[i] = q * 128;
            aswz[i] = (q >> 1) & 7;
        }
        f16x8 af[2][2], bf[2][2];
        auto load_frags = [&](int buf, int ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[buf][i] = *reinterpret_cast<const f16x8*>(pbuf + abase[i] + (((ks * 2 + hi) ^ aswz[i]) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bf[buf][j] = *reinterpret_cast<const f16x8*>(sB + lds_off(wn * 64 + j * 32 + l31, ks * 2 + hi));
        };
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) load_frags((ks + 1) & 1, ks + 1);
            if (ks == 1) {                                     // this iteration's group, behind the first MFMAs
                if (it + 2 < nt) issue_w(it + 2);
                if (tap < 7 && more) {
#pragma unroll
                    for (int j = 0; j < PT; ++j) issue_piece(ch + 1, tap * PT + j);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
        }
        if (++stage == S) stage = 0;
        if (++kx == 3) { kx = 0; ++ky; }
        if (++tap == 9) { tap = 0; ky = 0; ++ch; }
    }
    __syncthreads();
    tile_epilogue<BM, BN, WGM, WGN>(p, smem, acc, m0, n0, z);
}

// ------------------------------------------------------------------------------------------------ host side
struct Plan { int bm, bn, splits, fat; };

// Development options: storygen_amd/csrc/common.h SgOptions (set through sg_debug_set_option; never from the environment).
// conv_patch is OFF by default: measured on MI355X (round 2, tools/exp_feed.py) conv_patch_kernel ties the gathering kernel on
// every convolution of the step (47.2 vs 47.2 us at 64x64 320->320) although it moves 2-3x fewer bytes through the L1 — which is
// what showed that the mainloop is not bound by operand bytes.
struct TuneView {
    SgOptions& o = sg_options();
    int& bm = o.tile_m; int& bn = o.tile_n; int& no_pipe = o.no_pipe; int& no_split = o.no_split; int& stages = o.stages;
    int& no_nmajor = o.no_nmajor; int& late_issue = o.late_issue; int& no_frag_prefetch = o.no_frag_prefetch; int& fat = o.fat;
    int& conv_patch = o.conv_patch; int& spread = o.spread; int& pingpong = o.pingpong;
};
static const TuneView g_tune;

// Cost model (cycles at ~2.4 GHz).  Measured on MI355X (tools/bench_gemm.py): a CU pulls operand slabs from L2 into
// LDS at ~18.5 B/cycle however many waves ask (L1 miss-level parallelism x L2 latency), so a launch is bound by
//   t_bw   = (blocks a CU must run) x (bytes one block streams) / 18.5      — total bytes over the ACTIVE CUs, or by
//   t_mfma = (waves per SIMD) x slabs x 512                                  — 16 MFMAs of 32 cycles per 64-deep slab,
// plus a fixed prologue/epilogue.  Splitting K does not add operand bytes but multiplies the CUs that share them,
// which is what small-M layers need; it costs a second launch that re-reads the fp32 partial tiles.
// patch_w > 0: plan for conv_patch_kernel on an image of width patch_w and hw pixels per image (tiles are whole rows; the A
// side streams (bm / w + 2)(w + 2) pixels per 64-channel chunk instead of 9 bm; K is split in whole chunks of 9 slabs).
Plan choose_plan(int M, int N, int KT, int force_split, int max_ws_split, bool pipe, int hint_bm, int hint_bn, int hint_waves,
                 int patch_w = 0, int patch_hw = 0) {
    static const int cand_pipe[6][2] = {{256, 128}, {128, 128}, {256, 64}, {128, 64}, {64, 128}, {64, 64}};
    static const int cand_gen[3][2] = {{128, 128}, {128, 64}, {64, 64}};
    static const int split_opts[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
    const int ncand = pipe ? 6 : 3;
    const double CUS = 256.0, BW = pipe ? 18.5 : 12.0;
    Plan best{64, 64, 1, 0};
    double best_cost = 1e300;
    for (int ci = 0; ci < ncand; ++ci) {
        const int bm = pipe ? cand_pipe[ci][0] : cand_gen[ci][0], bn = pipe ? cand_pipe[ci][1] : cand_gen[ci][1];
        if (g_tune.bm && (bm != g_tune.bm || bn != g_tune.bn)) continue;
        if (!g_tune.bm && hint_bm && (bm != hint_bm || bn != hint_bn)) continue;
        if (patch_w && (bm % patch_w || patch_hw % bm)) continue;
        const double a_rows = patch_w ? (bm / patch_w + 2) * (patch_w + 2) / 9.0 : bm;    // A rows streamed per 64-deep slab
        const long tiles = (long)sg_cdiv(M, bm) * sg_cdiv(N, bn);
        const double waves_per_block = pipe ? (bm / 64) * (bn / 64) : 4.0;
        const double mfma_per_slab = pipe ? 512.0 : 512.0 * (bm / 64.0) * (bn / 64.0) / 4.0;
        for (int s : split_opts) {
            if (force_split > 0 && s != force_split) continue;
            if (force_split <= 0 && s > 1 && (s > max_ws_split || KT / s < 2 || g_tune.no_split)) continue;
            if (s > KT) continue;
            if (patch_w && s > KT / 9) continue;
            const double blocks = (double)tiles * s;
            const double slabs = patch_w ? 9.0 * sg_cdiv(KT / 9, s) : sg_cdiv(KT, s);
            const double blocks_per_cu = sg_cdiv((long)blocks, (long)CUS);
            const double t_bw = blocks_per_cu * slabs * (a_rows + bn) * 128.0 / BW;
            const double waves_per_simd = sg_cdiv((long)(blocks * waves_per_block), (long)(CUS * 4));
            const double t_mfma = waves_per_simd * slabs * mfma_per_slab;
            double cost = (t_bw > t_mfma ? t_bw : t_mfma) + 2500.0 + blocks_per_cu * (bm * bn / 16.0);
            if (s > 1) cost += 5000.0 + (double)M * N * 4.0 * (s + 1) / 1500.0;   // second launch + partial tiles
            if (cost < best_cost) { best_cost = cost; best = Plan{bm, bn, s, 0}; }
        }
    }
    if (best_cost == 1e300) {
        if (patch_w) return Plan{0, 0, 0, 0};                 // no eligible tile: the caller falls back to the gathering kernel
        best = Plan{64, 64, force_split > KT ? KT : (force_split > 0 ? force_split : 1), 0};
    }
    // "fat" waves (128 x 64 per wave): 256x128 with 4 waves, 128x128 with 2 — on request (tile_waves hint) or SG_FAT=1
    const bool can_fat = pipe && ((best.bm == 256 && best.bn == 128) || (best.bm == 128 && best.bn == 128));
    const int fat_waves = best.bm == 256 ? 4 : 2;
    if (can_fat && (hint_waves == fat_waves || (hint_waves == 0 && g_tune.fat))) best.fat = 1;
    return best;
}

template <int WGM, int WGN, bool CONV>
void launch_pipe_fat(const MmaParams& p, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((mma_pipe_kernel<WGM, WGN, 3, CONV, true, true, 4, 2>), grid, dim3(64 * WGM * WGN), 0, st, p);
}

template <int WGM, int WGN, bool CONV>
void launch_pipe(const MmaParams& p, dim3 grid, hipStream_t st, int stages) {
    const bool late = g_tune.late_issue != 0, pref = g_tune.no_frag_prefetch == 0;
    const dim3 block(64 * WGM * WGN);
    if (p.prof) {
        if (g_tune.spread == 2) hipLaunchKernelGGL((mma_pipe_prof_kernel<WGM, WGN, CONV, 2>), grid, block, 0, st, p);
        else if (g_tune.spread == 1) hipLaunchKernelGGL((mma_pipe_prof_kernel<WGM, WGN, CONV, 1>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((mma_pipe_prof_kernel<WGM, WGN, CONV, 0>), grid, block, 0, st, p);
        return;
    }
    if constexpr ((WGM + WGN) * 64 * 128 * 4 <= 160 * 1024) {      // the two 2-stage rings fit: every tile but 256 x 128
        if (g_tune.pingpong) {
            hipLaunchKernelGGL((mma_pp_kernel<WGM, WGN, CONV>), grid, dim3(128 * WGM * WGN), 0, st, p);
            return;
        }
    }
    if (g_tune.spread && stages == 3) {
        if (g_tune.spread == 2) hipLaunchKernelGGL((mma_pipe_spread_kernel<WGM, WGN, CONV, 2>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((mma_pipe_spread_kernel<WGM, WGN, CONV, 1>), grid, block, 0, st, p);
        return;
    }
    if constexpr ((WGM + WGN) * 64 * 128 * 4 <= 128 * 1024) {   // 4-deep ring where it fits (tiles up to 128 x 128)
        if (stages == 4) {
            hipLaunchKernelGGL((mma_pipe_kernel<WGM, WGN, 4, CONV, true, true>), grid, block, 0, st, p);
            return;
        }
    }
    if (stages == 2) hipLaunchKernelGGL((mma_pipe_kernel<WGM, WGN, 2, CONV, false, true>), grid, block, 0, st, p);
    else if (late && pref) hipLaunchKernelGGL((mma_pipe_kernel<WGM, WGN, 3, CONV, true, true>), grid, block, 0, st, p);
    else if (late) hipLaunchKernelGGL((mma_pipe_kernel<WGM, WGN, 3, CONV, true, false>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((mma_pipe_kernel<WGM, WGN, 3, CONV, false, true>), grid, block, 0, st, p);
}

thread_local unsigned long long* g_prof = nullptr;     // set by sg_debug_*_anatomy around one launch
thread_local int g_query_rows = 0;                     // result of a stats query (rows per partial = the tile height), 0 = none
thread_local bool g_stats_query = false;               // sg_*_stats_tile_rows: plan only, report eligibility instead of failing

// GroupNorm statistics from the epilogue (MmaParams::stats) need whole tiles inside one image, the fused (non split-K) linear
// epilogue, and N % 8 == 0.  A launch that was asked for them but cannot deliver fails (the caller asks sg_*_stats_tile_rows first).
int check_stats(MmaParams& p, int bm, const char* name) {
    if (!p.stats) return SG_OK;
    const bool ok = p.splits == 1 && p.mode == SG_EPI_LINEAR && p.stats_batch_rows > 0 && p.stats_batch_rows % bm == 0 &&
                    p.M % p.stats_batch_rows == 0;
    if (ok) return SG_OK;
    if (g_stats_query) { p.stats = nullptr; return SG_OK; }
    return sg_set_error(SG_EINVAL, "%s: epilogue statistics need a %d-row tile that divides the %d rows of an image and no split-K "
                        "(split %d): query sg_*_stats_tile_rows first", name, bm, p.stats_batch_rows, p.splits);
}

// Decomposition of one problem: tile shape, K split, tile order; fills the corresponding fields of p.  `pipe` = the LDS-DMA
// kernel applies (no load needs a predicate: K % 64 == 0; conv input zero-bordered), else the register-staged kernel.
template <bool CONV>
int plan_mma(MmaParams& p, int force_split, int hint_bm, int hint_bn, int hint_waves, void* ws, size_t ws_bytes, const char* name,
             Plan& pl, bool& pipe) {
    p.KT = sg_cdiv(p.K, BK);
    p.prof = g_prof;
    pipe = (p.K % BK == 0) && (!CONV || p.padded) && !g_tune.no_pipe;
    const size_t per_split = (size_t)p.M * p.N * 4;
    const int max_ws_split = ws ? (int)(ws_bytes / per_split > 64 ? 64 : ws_bytes / per_split) : 1;
    pl = choose_plan(p.M, p.N, p.KT, force_split, max_ws_split, pipe, hint_bm, hint_bn, hint_waves);
    if (pl.splits > 1) {
        const size_t need = per_split * pl.splits;
        if (ws == nullptr || ws_bytes < need)
            return sg_set_error(SG_EINVAL, "%s: split_k=%d needs %zu workspace bytes, got %zu", name, pl.splits, need,
                                ws_bytes);
    }
    p.ws = reinterpret_cast<float*>(ws);
    p.splits = pl.splits;
    p.kt_per_split = sg_cdiv(p.KT, pl.splits);
    p.tiles_m = sg_cdiv(p.M, pl.bm);
    p.tiles_n = sg_cdiv(p.N, pl.bn);
    // Each XCD has a private L2 and consecutive tile ids share one (xcd_remap): let them share the LARGER operand panel, so
    // that it is fetched from HBM / Infinity Cache by one XCD instead of by all that own a tile of it.  At the 16x16 and
    // 8x8 latent levels the weights (up to 59 MB per layer) dwarf the activations: walk M first there.
    const double a_bytes = CONV ? 2.0 * p.M * (p.K / 9) * (p.stride == 1 && !p.ups ? 1.0 : (p.ups ? 0.25 : 4.0)) : 2.0 * p.M * p.K;
    const double w_bytes = 2.0 * p.N * p.K;
    p.n_major = (w_bytes > a_bytes && !g_tune.no_nmajor) ? 1 : 0;
    return check_stats(p, pl.bm, name);
}

// second pass of a split-K launch: partial tiles -> epilogue
int launch_reduce(const MmaParams& p, hipStream_t st) {
    if (p.splits <= 1) return SG_OK;
    const long items = (long)p.M * (p.N / (p.mode == SG_EPI_GEGLU ? 16 : 8));
    const int blocks = (int)min((long)4096, (items + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    SG_CHECK_LAUNCH("splitk_reduce");
    return SG_OK;
}

template <int WGM, int WGN>
void launch_patch(const MmaParams& p, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((conv_patch_kernel<WGM, WGN>), grid, dim3(64 * WGM * WGN), 0, st, p);
}

// conv_patch_kernel when the problem allows it (stride 1, no upsampling, zero-bordered input, image width <= 64 dividing a tile
// that divides the image).  Returns 1 if launched, 0 if not applicable, < 0 on error.
int try_launch_conv_patch(MmaParams& p, int force_split, int hint_bm, int hint_bn, void* ws, size_t ws_bytes, hipStream_t st,
                          const char* name) {
    if (!g_tune.conv_patch || g_tune.no_pipe || !p.padded || p.stride != 1 || p.ups || p.Wd > 64 || p.Wd < 4) return 0;
    p.KT = 9 * p.cpt;
    const size_t per_split = (size_t)p.M * p.N * 4;
    const int max_ws_split = ws ? (int)(ws_bytes / per_split > 64 ? 64 : ws_bytes / per_split) : 1;
    Plan pl = choose_plan(p.M, p.N, p.KT, force_split, max_ws_split, true, hint_bm, hint_bn, 0, p.Wd, p.Ho * p.Wo);
    if (pl.bm == 0) return 0;
    const int cps = sg_cdiv(p.cpt, pl.splits);                // whole channel chunks per slice
    pl.splits = sg_cdiv(p.cpt, cps);
    if (pl.splits > 1 && (ws == nullptr || ws_bytes < per_split * pl.splits)) {
        if (force_split > 1) return sg_set_error(SG_EINVAL, "%s: split_k=%d does not fit the workspace", name, pl.splits);
        return 0;
    }
    p.ws = reinterpret_cast<float*>(ws);
    p.splits = pl.splits;
    p.kt_per_split = 9 * cps;
    p.tiles_m = p.M / pl.bm;
    p.tiles_n = sg_cdiv(p.N, pl.bn);
    const double a_bytes = 2.0 * p.M * (p.K / 9), w_bytes = 2.0 * p.N * p.K;
    p.n_major = (w_bytes > a_bytes && !g_tune.no_nmajor) ? 1 : 0;
    if (int rc = check_stats(p, pl.bm, name)) return rc;
    if (g_stats_query) {
        g_query_rows = p.stats ? pl.bm : 0;
        return 1;
    }
    dim3 grid(p.tiles_m * p.tiles_n * pl.splits);
    if (pl.bm == 256 && pl.bn == 128) launch_patch<4, 2>(p, grid, st);
    else if (pl.bm == 128 && pl.bn == 128) launch_patch<2, 2>(p, grid, st);
    else if (pl.bm == 256 && pl.bn == 64) launch_patch<4, 1>(p, grid, st);
    else if (pl.bm == 128 && pl.bn == 64) launch_patch<2, 1>(p, grid, st);
    else if (pl.bm == 64 && pl.bn == 128) launch_patch<1, 2>(p, grid, st);
    else launch_patch<1, 1>(p, grid, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return sg_set_error(SG_ELAUNCH, "%s: %s", name, hipGetErrorString(e));
    if (int rc = launch_reduce(p, st)) return rc;
    return 1;
}

template <bool CONV>
int launch_mma(MmaParams& p, int force_split, int hint_bm, int hint_bn, int hint_waves, void* ws, size_t ws_bytes, hipStream_t st, const char* name) {
    if constexpr (CONV) {
        if (hint_waves == 0) {
            const int rc = try_launch_conv_patch(p, force_split, hint_bm, hint_bn, ws, ws_bytes, st, name);
            if (rc != 0) return rc < 0 ? rc : SG_OK;
        }
    }
    Plan pl;
    bool pipe;
    if (int rc = plan_mma<CONV>(p, force_split, hint_bm, hint_bn, hint_waves, ws, ws_bytes, name, pl, pipe)) return rc;
    if (g_stats_query) {
        g_query_rows = p.stats ? pl.bm : 0;
        return SG_OK;
    }
    dim3 grid(p.tiles_m * p.tiles_n * pl.splits);
    // ring depth: 3 stages (deeper prefetch) unless overridden; SG_STAGES=2 halves... see DESIGN.md §5.2
    const int stages = (g_tune.stages == 2 || g_tune.stages == 4) ? g_tune.stages : 3;
    if (pipe && pl.fat) {
        if (pl.bm == 256) launch_pipe_fat<2, 2, CONV>(p, grid, st);
        else launch_pipe_fat<1, 2, CONV>(p, grid, st);
    } else if (pipe) {
        if (pl.bm == 256 && pl.bn == 128) launch_pipe<4, 2, CONV>(p, grid, st, stages);
        else if (pl.bm == 128 && pl.bn == 128) launch_pipe<2, 2, CONV>(p, grid, st, stages);
        else if (pl.bm == 256 && pl.bn == 64) launch_pipe<4, 1, CONV>(p, grid, st, stages);
        else if (pl.bm == 128 && pl.bn == 64) launch_pipe<2, 1, CONV>(p, grid, st, stages);
        else if (pl.bm == 64 && pl.bn == 128) launch_pipe<1, 2, CONV>(p, grid, st, stages);
        else launch_pipe<1, 1, CONV>(p, grid, st, stages);
    } else {
        dim3 block(256);
        if (pl.bm == 128 && pl.bn == 128) hipLaunchKernelGGL((mma_kernel<128, 128, CONV>), grid, block, 0, st, p);
        else if (pl.bm == 128 && pl.bn == 64) hipLaunchKernelGGL((mma_kernel<128, 64, CONV>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((mma_kernel<64, 64, CONV>), grid, block, 0, st, p);
    }
    SG_CHECK_LAUNCH(name);
    return launch_reduce(p, st);
}

int check_out_res(const char* who, int flags, const void* C, int64_t ldc, const void* C2, int64_t ldc2, const void* res1,
                  int64_t ldr1, const void* res2, int64_t ldr2, int n_out) {
    SG_REQUIRE((flags & ~(SG_F_OUT_F32 | SG_F_RES1_F32 | SG_F_RES2_F32)) == 0, "%s: unknown flag bits 0x%x", who, flags);
    SG_REQUIRE(sg_aligned16(C) && ldc % 8 == 0 && ldc >= n_out, "%s: output alignment / ld", who);
    SG_REQUIRE(!C2 || (sg_aligned16(C2) && ldc2 % 8 == 0 && ldc2 >= n_out), "%s: second output alignment / ld", who);
    SG_REQUIRE(!res1 || (sg_aligned16(res1) && ldr1 % 8 == 0), "%s: res1 alignment", who);
    SG_REQUIRE(!res2 || (sg_aligned16(res2) && ldr2 % 8 == 0), "%s: res2 alignment", who);
    return SG_OK;
}

int check_tile_hint(const char* who, int bm, int bn, int waves) {
    if (waves != 0 && !((bm == 256 && bn == 128 && (waves == 4 || waves == 8)) || (bm == 128 && bn == 128 && (waves == 2 || waves == 4))))
        return sg_set_error(SG_EINVAL, "%s: tile_waves=%d is not available for tile %dx%d", who, waves, bm, bn);
    if (bm == 0 && bn == 0) return SG_OK;
    static const int ok[6][2] = {{256, 128}, {128, 128}, {256, 64}, {128, 64}, {64, 128}, {64, 64}};
    for (auto& t : ok)
        if (t[0] == bm && t[1] == bn) return SG_OK;
    return sg_set_error(SG_EINVAL, "%s: unsupported tile hint %dx%d", who, bm, bn);
}

}  // namespace

extern "C" size_t sg_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k) {
    const int s = split_k > 0 ? split_k : MAX_AUTO_SPLIT;
    return s > 1 ? (size_t)M * (size_t)N * 4u * (size_t)s : 0;
}

namespace {
// Validates a GEMM descriptor and translates it into kernel parameters.
int gemm_params(const sg_gemm_desc* d, MmaParams& p, const char* who) {
    SG_REQUIRE(d != nullptr, "%s: null descriptor", who);
    SG_REQUIRE(d->A && d->W && d->C, "%s: null A/W/C", who);
    SG_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "%s: bad shape M=%d N=%d K=%d", who, d->M, d->N, d->K);
    SG_REQUIRE(d->K % 8 == 0 && d->N % 8 == 0, "%s: K (%d) and N (%d) must be multiples of 8", who, d->K, d->N);
    SG_REQUIRE(d->lda % 8 == 0 && d->ldw % 8 == 0, "%s: lda/ldw must be multiples of 8", who);
    SG_REQUIRE(d->lda >= d->K && d->ldw >= d->K, "%s: lda/ldw smaller than K", who);
    SG_REQUIRE(sg_aligned16(d->A) && sg_aligned16(d->W), "%s: A/W must be 16-byte aligned", who);
    SG_REQUIRE(d->epilogue == SG_EPI_LINEAR || d->epilogue == SG_EPI_GEGLU, "%s: unknown epilogue %d", who, d->epilogue);
    int n_out = d->N;
    if (d->epilogue == SG_EPI_GEGLU) {
        SG_REQUIRE(d->N % 64 == 0, "%s: GEGLU needs N %% 64 == 0 (got %d)", who, d->N);
        SG_REQUIRE(!d->rowbias && !d->res1 && !d->res2, "%s: GEGLU epilogue takes bias only", who);
        n_out = d->N / 2;
    }
    if (int rc = check_out_res(who, d->flags, d->C, d->ldc, d->C2, d->ldc2, d->res1, d->ldr1, d->res2, d->ldr2, n_out))
        return rc;
    SG_REQUIRE(!d->bias || sg_aligned16(d->bias), "%s: bias must be 16-byte aligned", who);
    SG_REQUIRE(!d->rowbias || (sg_aligned16(d->rowbias) && d->rowbias_ld % 4 == 0 && d->rows_per_batch >= 1),
               "%s: rowbias alignment / rows_per_batch", who);
    SG_REQUIRE(d->split_k >= 0 && d->split_k <= 64, "%s: bad split_k %d", who, d->split_k);
    SG_REQUIRE(!d->workspace || (sg_aligned16(d->workspace)), "%s: workspace alignment", who);
    SG_REQUIRE((int64_t)d->M * d->lda < (1ll << 32) && (int64_t)d->N * d->ldw < (1ll << 32),
               "%s: operands larger than 2^32 elements are not supported (32-bit DMA offsets)", who);
    p = MmaParams{};
    p.A = reinterpret_cast<const f16*>(d->A); p.lda = d->lda;
    p.W = reinterpret_cast<const f16*>(d->W); p.ldw = d->ldw;
    p.C = d->C; p.ldc = d->ldc;
    p.C2 = reinterpret_cast<f16*>(d->C2); p.ldc2 = d->ldc2;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.mode = d->epilogue; p.flags = d->flags;
    p.bias = reinterpret_cast<const f16*>(d->bias);
    p.rowbias = d->rowbias; p.rowbias_ld = d->rowbias_ld; p.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : 1;
    p.res1 = d->res1; p.ldr1 = d->ldr1;
    p.res2 = d->res2; p.ldr2 = d->ldr2;
    SG_REQUIRE(!d->stats || (sg_aligned16(d->stats) && d->stats_batch_rows > 0), "%s: stats alignment / stats_batch_rows", who);
    p.stats = d->stats; p.stats_batch_rows = d->stats_batch_rows;
    return check_tile_hint(who, d->tile_m, d->tile_n, d->tile_waves);
}

template <int WGM, int WGN>
void launch_pair(const MmaPair& pp, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((mma_pipe_pair_kernel<WGM, WGN>), grid, dim3(64 * WGM * WGN), 0, st, pp);
}
}  // namespace

extern "C" int sg_gemm_f16(const sg_gemm_desc* d, sg_stream_t stream) {
    MmaParams p;
    if (int rc = gemm_params(d, p, "sg_gemm_f16")) return rc;
    return launch_mma<false>(p, d->split_k, d->tile_m, d->tile_n, d->tile_waves, d->workspace, d->workspace_bytes, (hipStream_t)stream, "sg_gemm_f16");
}

extern "C" int sg_gemm_pair_f16(const sg_gemm_desc* d0, const sg_gemm_desc* d1, sg_stream_t stream) {
    MmaPair pp;
    if (int rc = gemm_params(d0, pp.p0, "sg_gemm_pair_f16[0]")) return rc;
    if (int rc = gemm_params(d1, pp.p1, "sg_gemm_pair_f16[1]")) return rc;
    SG_REQUIRE(!(d0->workspace && d1->workspace) ||
               (reinterpret_cast<char*>(d0->workspace) + d0->workspace_bytes <= reinterpret_cast<char*>(d1->workspace) ||
                reinterpret_cast<char*>(d1->workspace) + d1->workspace_bytes <= reinterpret_cast<char*>(d0->workspace)),
               "sg_gemm_pair_f16: the two problems run concurrently and need disjoint workspaces");
    hipStream_t st = (hipStream_t)stream;
    Plan pl0, pl1;
    bool pipe0, pipe1;
    if (int rc = plan_mma<false>(pp.p0, d0->split_k, d0->tile_m, d0->tile_n, 0, d0->workspace, d0->workspace_bytes, "sg_gemm_pair_f16[0]", pl0, pipe0)) return rc;
    // one kernel instantiation serves both problems: the second one is planned on the first one's tile shape
    if (int rc = plan_mma<false>(pp.p1, d1->split_k, pl0.bm, pl0.bn, 0, d1->workspace, d1->workspace_bytes, "sg_gemm_pair_f16[1]", pl1, pipe1)) return rc;
    if (!pipe0 || !pipe1 || pl1.bm != pl0.bm || pl1.bn != pl0.bn) {        // not pairable (K % 64, forced tile): two launches
        if (int rc = launch_mma<false>(pp.p0, d0->split_k, d0->tile_m, d0->tile_n, d0->tile_waves, d0->workspace, d0->workspace_bytes, st, "sg_gemm_pair_f16[0]")) return rc;
        return launch_mma<false>(pp.p1, d1->split_k, d1->tile_m, d1->tile_n, d1->tile_waves, d1->workspace, d1->workspace_bytes, st, "sg_gemm_pair_f16[1]");
    }
    const int g0 = pp.p0.tiles_m * pp.p0.tiles_n * pp.p0.splits, g1 = pp.p1.tiles_m * pp.p1.tiles_n * pp.p1.splits;
    // grid.x is a multiple of 8 so that block (x, 1) sits on XCD x % 8 like block (x, 0): xcd_remap keeps its meaning
    dim3 grid(((g0 > g1 ? g0 : g1) + 7) & ~7, 2);
    if (pl0.bm == 256 && pl0.bn == 128) launch_pair<4, 2>(pp, grid, st);
    else if (pl0.bm == 128 && pl0.bn == 128) launch_pair<2, 2>(pp, grid, st);
    else if (pl0.bm == 256 && pl0.bn == 64) launch_pair<4, 1>(pp, grid, st);
    else if (pl0.bm == 128 && pl0.bn == 64) launch_pair<2, 1>(pp, grid, st);
    else if (pl0.bm == 64 && pl0.bn == 128) launch_pair<1, 2>(pp, grid, st);
    else launch_pair<1, 1>(pp, grid, st);
    SG_CHECK_LAUNCH("sg_gemm_pair_f16");
    if (int rc = launch_reduce(pp.p0, st)) return rc;
    return launch_reduce(pp.p1, st);
}

extern "C" int sg_conv3x3_nhwc_f16(const sg_conv3x3_desc* d, sg_stream_t stream) {
    SG_REQUIRE(d != nullptr, "sg_conv3x3: null descriptor");
    SG_REQUIRE(d->x && d->w && d->y, "sg_conv3x3: null x/w/y");
    SG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "sg_conv3x3: bad shape");
    SG_REQUIRE(d->Cin % 64 == 0, "sg_conv3x3: Cin (%d) must be a multiple of 64", d->Cin);
    SG_REQUIRE(d->Cout % 8 == 0, "sg_conv3x3: Cout (%d) must be a multiple of 8", d->Cout);
    SG_REQUIRE(d->stride == 1 || d->stride == 2, "sg_conv3x3: stride must be 1 or 2");
    SG_REQUIRE(d->upsample2x == 0 || (d->upsample2x == 1 && d->stride == 1), "sg_conv3x3: upsample2x needs stride 1");
    SG_REQUIRE(d->x_padded == 0 || d->x_padded == 1, "sg_conv3x3: x_padded must be 0 or 1");
    SG_REQUIRE(d->ldx % 8 == 0 && d->ldx >= d->Cin, "sg_conv3x3: bad ldx");
    SG_REQUIRE(sg_aligned16(d->x) && sg_aligned16(d->w), "sg_conv3x3: x/w must be 16-byte aligned");
    if (int rc = check_out_res("sg_conv3x3", d->flags, d->y, d->ldy, nullptr, 0, d->res1, d->ldr1, nullptr, 0, d->Cout)) return rc;
    SG_REQUIRE(!d->bias || sg_aligned16(d->bias), "sg_conv3x3: bias alignment");
    SG_REQUIRE(!d->rowbias || (sg_aligned16(d->rowbias) && d->rowbias_ld % 4 == 0), "sg_conv3x3: rowbias alignment");
    SG_REQUIRE(d->split_k >= 0 && d->split_k <= 64, "sg_conv3x3: bad split_k %d", d->split_k);
    SG_REQUIRE(!d->workspace || sg_aligned16(d->workspace), "sg_conv3x3: workspace alignment");
    SG_REQUIRE((int64_t)d->B * (d->H + 2) * (d->W + 2) * d->ldx < (1ll << 32) && (int64_t)d->Cout * 9 * d->Cin < (1ll << 32),
               "sg_conv3x3: operands larger than 2^32 elements are not supported (32-bit DMA offsets)");
    SG_REQUIRE((int64_t)(d->W + 2) * d->ldx < (1 << 24), "sg_conv3x3: input row pitch must be below 2^24 elements");
    const int hin = d->H << d->upsample2x, win = d->W << d->upsample2x;
    const int Ho = (hin + 2 - 3) / d->stride + 1, Wo = (win + 2 - 3) / d->stride + 1;
    MmaParams p{};
    p.A = reinterpret_cast<const f16*>(d->x); p.lda = d->ldx;
    p.W = reinterpret_cast<const f16*>(d->w); p.ldw = 9L * d->Cin;
    p.C = d->y; p.ldc = d->ldy;
    p.M = d->B * Ho * Wo; p.N = d->Cout; p.K = 9 * d->Cin;
    p.H = d->H; p.Wd = d->W; p.Ho = Ho; p.Wo = Wo; p.cpt = d->Cin / 64; p.stride = d->stride; p.ups = d->upsample2x;
    p.padded = d->x_padded;
    p.mode = SG_EPI_LINEAR; p.flags = d->flags;
    p.bias = reinterpret_cast<const f16*>(d->bias);
    p.rowbias = d->rowbias; p.rowbias_ld = d->rowbias_ld; p.rows_per_batch = Ho * Wo;
    p.res1 = d->res1; p.ldr1 = d->ldr1;
    SG_REQUIRE(!d->stats || sg_aligned16(d->stats), "sg_conv3x3: stats alignment");
    p.stats = d->stats; p.stats_batch_rows = Ho * Wo;
    if (int rc = check_tile_hint("sg_conv3x3", d->tile_m, d->tile_n, d->tile_waves)) return rc;
    return launch_mma<true>(p, d->split_k, d->tile_m, d->tile_n, d->tile_waves, d->workspace, d->workspace_bytes, (hipStream_t)stream, "sg_conv3x3_nhwc_f16");
}

// Would a launch with this descriptor emit epilogue statistics, and with which tile height?  (Plans the launch exactly as
// sg_gemm_f16 / sg_conv3x3_nhwc_f16 would — the plan is a pure function of the descriptor and the development options — without
// launching.)  Returns the rows per partial (> 0), 0 when the launch cannot emit them, < 0 on an invalid descriptor.
extern "C" int sg_gemm_stats_tile_rows(const sg_gemm_desc* d) {
    SG_REQUIRE(d && d->stats, "sg_gemm_stats_tile_rows: descriptor with a stats buffer required");
    g_stats_query = true; g_query_rows = 0;
    const int rc = sg_gemm_f16(d, nullptr);
    g_stats_query = false;
    return rc ? rc : g_query_rows;
}

extern "C" int sg_conv3x3_stats_tile_rows(const sg_conv3x3_desc* d) {
    SG_REQUIRE(d && d->stats, "sg_conv3x3_stats_tile_rows: descriptor with a stats buffer required");
    g_stats_query = true; g_query_rows = 0;
    const int rc = sg_conv3x3_nhwc_f16(d, nullptr);
    g_stats_query = false;
    return rc ? rc : g_query_rows;
}

// ------------------------------------------------------------------------------------------------ diagnostics
// Mainloop anatomy: the same launch as sg_gemm_f16 / sg_conv3x3_nhwc_f16 through the instrumented instantiation of the pipelined
// kernel; prof receives 10 uint64 per wave ([block][wave][10], see mma_pipe_body).  Needs the LDS-DMA path, no fat waves, S = 3.
extern "C" int sg_debug_gemm_anatomy(const sg_gemm_desc* d, void* prof, size_t prof_bytes, sg_stream_t stream) {
    SG_REQUIRE(d && prof && prof_bytes >= (size_t)8 * 10 * 8 * 65536 / 64, "sg_debug_gemm_anatomy: need a profile buffer (>= 80 B per wave)");
    g_prof = reinterpret_cast<unsigned long long*>(prof);
    const int rc = sg_gemm_f16(d, stream);
    g_prof = nullptr;
    return rc;
}

extern "C" int sg_debug_conv_anatomy(const sg_conv3x3_desc* d, void* prof, size_t prof_bytes, sg_stream_t stream) {
    SG_REQUIRE(d && prof && prof_bytes >= (size_t)8 * 10 * 8 * 65536 / 64, "sg_debug_conv_anatomy: need a profile buffer (>= 80 B per wave)");
    g_prof = reinterpret_cast<unsigned long long*>(prof);
    const int rc = sg_conv3x3_nhwc_f16(d, stream);
    g_prof = nullptr;
    return rc;
}

extern "C" int sg_debug_set_tile(int32_t bm, int32_t bn, int32_t no_pipe) {
    g_tune.bm = bm; g_tune.bn = bn; g_tune.no_pipe = no_pipe;
    return SG_OK;
}

namespace {
__global__ void debug_mfma_kernel(const f16* a, const f16* b, float* out) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    f16x8 af, bf;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        af[j] = a[l31 * 16 + hi * 8 + j];          // A[i = l31][k = hi*8 + j]
        bf[j] = b[(hi * 8 + j) * 32 + l31];        // B[k = hi*8 + j][n = l31]
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}
}  // namespace

// Probe for the block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, fp8 e4m3 operands, unit scales): one wave, raw
// per-lane operand bytes in, raw per-lane accumulator registers out.  tools/probe_mfma_f8.py uses it to establish the
// (lane, byte) -> (row / column, k) operand maps on the device before any fp8 attention kernel (BASELINE config 5) is written.
typedef int v8i_probe __attribute__((ext_vector_type(8)));
__global__ void debug_mfma_f8_kernel(const int* a, const int* b, float* out, int scale_a, int scale_b) {
    const int lane = threadIdx.x;
    v8i_probe A, B;
#pragma unroll
    for (int i = 0; i < 8; ++i) { A[i] = a[lane * 8 + i]; B[i] = b[lane * 8 + i]; }
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 0, 0, 0, scale_a, 0, scale_b);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

extern "C" int sg_debug_mfma_f8_32x32x64(const void* a, const void* b, float* out, int32_t scale_a, int32_t scale_b, sg_stream_t stream) {
    SG_REQUIRE(a && b && out, "sg_debug_mfma_f8: null pointer");
    hipLaunchKernelGGL(debug_mfma_f8_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const int*>(a),
                       reinterpret_cast<const int*>(b), out, scale_a, scale_b);
    SG_CHECK_LAUNCH("sg_debug_mfma_f8_32x32x64");
    return SG_OK;
}

extern "C" int sg_debug_mfma_32x32x16(const sg_half* a, const sg_half* b, float* out, sg_stream_t stream) {
    SG_REQUIRE(a && b && out, "sg_debug_mfma: null pointer");
    hipLaunchKernelGGL(debug_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const f16*>(a),
                       reinterpret_cast<const f16*>(b), out);
    SG_CHECK_LAUNCH("sg_debug_mfma_32x32x16");
    return SG_OK;
}
