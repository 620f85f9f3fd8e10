// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X, CDNA4), fp16 operands / fp32 accumulate.
//
// One mainloop family serves both entry points (sg_gemm_f16, sg_conv3x3_nhwc_f16): C[M,N] = A[M,K] . W[N,K]^T where,
// for the convolution, row m is an output pixel and the K axis enumerates (ky, kx, ci) — the A tile is *gathered*
// from the NHWC input (optional nearest-2x upsample and stride 2 folded into the index).
//
// Structure (CDNA4-first, see /opt/skills/guides/cdna_hip_programming.md §5):
//   * every wave owns a 64x64 output sub-tile = 2x2 accumulators of v_mfma_f32_32x32x16_f16; a workgroup is a WGM x WGN grid of
//     such waves (256x128, 128x128, 256x64, 128x64, 64x128 or 64x64 tiles), chosen per problem so that the chip stays busy.
//   * pipelined kernel (the fast path): K is walked in 64-deep slabs through a 3-stage LDS ring filled by LDS-DMA
//     (global_load_lds, 16 B per lane, no VGPR staging), counted vmcnt waits and ONE raw s_barrier per slab.  Round 4: the slab body
//     exists once per ring stage (the stage is a compile-time constant: fragment reads are per-lane base registers + immediates), and
//     the refill of the ring is placed by waves per SIMD — spread over the four k-steps on the tiles of <= 4 waves, a burst staggered
//     between the two waves of a SIMD on the 8-wave tile (VALU and DMA-issue cycles ADD to matrix cycles on a SIMD: HISTORY.md 5.2d).
//   * LDS rows are 128 B (64 halves); the 16-byte chunk c of row r lives at slot c ^ ((r>>1)&7): conflict-free for
//     the ds_read_b128 fragment reads.  LDS-DMA writes lane l of a wave to (wave-uniform base + 16 l), so the swizzle is
//     applied on the SOURCE side (the lane that owns slot s of row r fetches logical chunk s ^ ((r>>1)&7)) and again on the reads.
//   * the accumulators are computed TRANSPOSED (the weight fragment is the MFMA's A operand, the activation fragment its B
//     operand): every register quad of a lane is then 4 consecutive output columns of ONE row = one ds_write_b128 into a row-major
//     fp32 image.  Each wave transposes its own 64x64 sub-tile through its own 17 KB of LDS — ONE workgroup barrier after the
//     mainloop (the ring must be dead), none between the waves — and reads it back as row quads, so that every global access of
//     the epilogue (bias, fp32 residual, fp32 / fp16 outputs) is 4 rows x 256 contiguous bytes per wave instruction.  All loads of
//     an epilogue term are issued together (clamped addresses, no predicates): one memory round trip per term; the fp32 residual and
//     the bias are requested during the LAST K slab, so their HBM latency runs under that slab's MFMAs (round 3; replaces the
//     banded, LDS-staged epilogue with 2 barriers and one residual round trip per 32-row band).
//   * LayerNorm folded into the consuming GEMM (round 3; sg_gemm_desc.ln_*): producers also emit the fp16 copy of their output and
//     per-token (sum, M2) partials per 64-column block; consumers run on the raw copy with gamma-scaled weights and their epilogue
//     applies rstd (acc - mean c) + d — rows-are-tokens, columns-are-tokens (the transposed V^T product) and GEGLU forms.
//   * generic kernel (fallback): register-staged double buffer with zero-fill predicates, for K % 64 != 0 or an
//     unpadded convolution input.
//   * block ids are remapped so that consecutive tiles (same operand panel) run on the same XCD / L2; index arithmetic of the
//     prologue uses host-made multiply-shift divisors (no integer division on the device).
//   * small-M layers (16x16 / 8x8 latent levels at batch 3) are split along K into fp32 partial tiles; a second kernel
//     reduces them in slice order and applies the epilogue (deterministic, no atomics).  An in-launch reduction by the
//     last-arriving slice (agent-scope release / acquire + ticket counter) was built and measured in round 2: the release
//     fence behind 64 KB of freshly written partials costs more (+8..13 us per launch) than the kernel boundary it removes.
// Round-2 mainloop experiments (4-/2-stage rings, spread DMA issue on the 256x128 tile, ping-pong wave groups, LDS-resident conv
// patches, fat waves) were measured neutral or slower (HISTORY.md 5.2) and are no longer part of the library; `git log` has them.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <vector>

namespace {

constexpr int BK = 64;
constexpr int MAX_AUTO_SPLIT = 16;

// n / d for any 32-bit n by multiply-high and shifts (Granlund-Montgomery; made on the host by make_fastdiv)
struct FastDiv { unsigned mul, sh1, sh2; };

__host__ __device__ __forceinline__ unsigned fd_div(unsigned n, FastDiv f) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned t = __umulhi(n, f.mul);
#else
    const unsigned t = (unsigned)(((unsigned long long)n * f.mul) >> 32);
#endif
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

FastDiv make_fastdiv(unsigned d) {
    if (d <= 1) return FastDiv{0u, 0u, 0u};
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    const unsigned long long m = (((1ull << l) - d) << 32) / d + 1;
    return FastDiv{(unsigned)m, 1u, l - 1};
}

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
    int defer;                     // split-K: stop after the partial tiles (sg_conv3x3_desc.defer_reduce); the consumer reduces
    float* stats; int stats_batch_rows;   // optional GroupNorm partial statistics of the output (epi_finish), else nullptr
    // LayerNorm folded into this GEMM (sg_gemm_desc.ln_*): consumer side = per-token (mean, M2) partials of the raw operand,
    // c / d vectors of the folded weight; producer side = where to write the partials of THIS output
    const float* ln_stats; int ln_parts; const float* ln_c; const float* ln_d; int ln_mode; float ln_eps;
    float* ln_out;
    unsigned* ln_guard;            // sticky flags of the fold's two assumptions (sg_gemm_desc.ln_guard), or nullptr
    unsigned long long* prof;   // SG_BUILD_EXPERIMENTS (sg_debug_*_anatomy): per-wave cycle totals of the mainloop phases
    // tile order (tile_of_id): consecutive logical ids walk group_m row tiles, then move one column tile on; after all column tiles the
    // next group of row tiles.  group_m >= tiles_m: M first throughout (tiles sharing a weight panel are neighbours); group_m = 1: N first.
    int group_m;
    FastDiv fd_splits, fd_group_w /* group_m * tiles_n */, fd_group_m, fd_hw, fd_wo, fd_rpb, fd_cpt;
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Grouped tile order (a bijection of [0, tiles_m * tiles_n) onto the tile grid).  The ~32 workgroups that are resident on an XCD at one
// time are consecutive logical ids (xcd_remap) and move through K roughly in step, so what that XCD's L2 has to fetch is one operand
// panel per DISTINCT row tile and per DISTINCT column tile among them: gm row tiles x 32 / gm column tiles.  Walking all of M first
// (round 1-4 when the weights were the larger operand) puts up to 32 different activation panels and one or two weight panels into a
// wave of workgroups — for M 5120 x N 10240 x K 1280 that re-fetched the 13 MB of activations for every weight panel, 883 MB per launch
// against 92 MB algorithmic (profiles/traffic.json, round-5 mid build).  gm is chosen by plan_mma to minimise the bytes of one wave.
__host__ __device__ __forceinline__ void tile_of_id(unsigned lid, int tiles_m, int tiles_n, int gm, FastDiv fd_w, FastDiv fd_gm,
                                                    unsigned& tm, unsigned& tn) {
    const unsigned width = (unsigned)gm * (unsigned)tiles_n;
    const unsigned group = fd_div(lid, fd_w), r = lid - group * width;
    const unsigned first = group * (unsigned)gm;
    const unsigned gsize = min((unsigned)tiles_m - first, (unsigned)gm);
    // the last group of row tiles may be smaller than gm: a plain division there (once per workgroup)
    const unsigned q = gsize == (unsigned)gm ? fd_div(r, fd_gm) : r / gsize;
    tn = q;
    tm = first + (r - q * gsize);
}

// logical block id -> (m0, n0, K slice z).  logical id = tile * splits + slice: the K slices of a tile are consecutive ids
// (one XCD, see xcd_remap)
__device__ __forceinline__ void decode_block(const MmaParams& p, int BM, int BN, int& m0, int& n0, int& z) {
    const unsigned lid2 = (unsigned)xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n * p.splits);
    const unsigned lid = fd_div(lid2, p.fd_splits);
    z = (int)(lid2 - lid * p.splits);
    unsigned tm, tn;
    tile_of_id(lid, p.tiles_m, p.tiles_n, p.group_m, p.fd_group_w, p.fd_group_m, tm, tn);
    m0 = (int)tm * BM;
    n0 = (int)tn * BN;
}

// ------------------------------------------------------------------------------------------------------------
// Epilogue.  acc[i][j] = mfma(W fragment, A fragment) is the TRANSPOSED product: register r of lane (l31, hi) is
//   output row    i*32 + l31                      (of the wave's 64x64 sub-tile)
//   output column j*32 + 8*(r>>2) + 4*hi + (r&3)
// i.e. every register quad is 4 consecutive columns of one row: ONE ds_write_b128 into a row-major fp32 image.  Each wave
// transposes its own sub-tile through its own 17 KB of LDS (row pitch 68 floats: the quad writes of 8 consecutive rows cover the
// 32 banks exactly) — no barrier between the waves — and reads it back as 16 "row quads": instruction k gives lane l the 4
// consecutive columns 4 (l & 15) of row 4 k + (l >> 4), so that every global access of the epilogue is 4 rows x 256 contiguous
// bytes per wave instruction (the first version of this round kept one row per lane: no LDS, but 32 cache lines per
// instruction — the stores and residual loads then cost what the barriers of the old banded epilogue had).
constexpr int EPI_PITCH = 68;                         // floats per staged row
constexpr int EPI_XTR = 64 * EPI_PITCH;               // float offset of a wave's 1 KB extra area: [0,128) LayerNorm coefficients,
                                                      // [128,256) its 64 columns x 2 planes of GroupNorm statistics (or more LN rows)
constexpr int EPI_WAVE_BYTES = 64 * EPI_PITCH * 4 + 1024;    // 18432
constexpr int EPIQ_GROUP_BYTES = 64 * EPI_PITCH * 4 + 4 * 1024;     // epi_finish_q: one 64x64 fp32 image + 1 KB per wave of a 2x2 group

constexpr int LN_MAX_PARTS = 20;      // 64-column blocks per token: C <= 1280
// Prefetched epilogue operands (native clang vectors, not HIP's float4 class: as members of a struct, or as HIP vector classes,
// the arrays were demoted to scratch memory): pre[k] = row quad k of the fp32 residual, pbias = bias of the lane's 4 columns.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define EPI_PRE_DECL f32x4 pre_f[16]; u32x2 pre_bias
#define EPI_PRE_ARGS pre_f, pre_bias
#define EPI_PRE_PARAMS f32x4 (&pre)[16], u32x2& pbias

// Epilogue traffic is streamed once (outputs written, residuals read): with SG_EPI_NT the accesses carry the non-temporal hint, so
// they do not evict the operand panels (weights, im2col rows re-read by every tap and column tile) from the XCD's 4 MB L2 — the PMC
// pass of round 4 measured 150 MB of fabric traffic per 64x64 320->320 convolution against 58 MB of compulsory traffic.
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_stream(float* p, const f32x4& v) {
#ifdef SG_EPI_NT
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}
__device__ __forceinline__ void st_stream(f16* p, const uint2& v) {
    const u32x2v w = {v.x, v.y};
#ifdef SG_EPI_NT
    __builtin_nontemporal_store(w, reinterpret_cast<u32x2v*>(p));
#else
    *reinterpret_cast<u32x2v*>(p) = w;
#endif
}
__device__ __forceinline__ f32x4 ld_stream(const float* p) {
#ifdef SG_EPI_NT
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
    return *reinterpret_cast<const f32x4*>(p);
#endif
}

// Everything the fused linear epilogue needs from memory that does not depend on the accumulators is requested early — during the
// last K slab — into registers: the fp32 residual (the UNet's residual stream) in the row-quad layout (16 B per item; rowq = first
// row of the wave's sub-tile + (lane >> 4), colq = first column + 4 (lane & 15)) and the bias.  Rows / columns beyond the problem
// are clamped, not predicated: unconditional loads issue back to back (a branch per load makes the compiler wait for each one
// where it is issued).
__device__ __forceinline__ bool epi_prefetches(const MmaParams& p) {
    return p.res1 != nullptr && (p.flags & SG_F_RES1_F32) && p.splits == 1 && p.mode == SG_EPI_LINEAR;
}

// (wrows = rows per wave: 64, or 128 for the 128x64-per-wave tiles, which fetch the bias only)
__device__ __forceinline__ void epi_prefetch(const MmaParams& p, int m0, int n0, int wm, int wn, int lane, EPI_PRE_PARAMS, int wrows = 64, int npre = 16) {
    if (p.splits > 1 || p.mode != SG_EPI_LINEAR) return;
    const int rowq = m0 + wm * wrows + (lane >> 4), cq = min(n0 + wn * 64 + 4 * (lane & 15), p.N - 4);
    if (p.bias) pbias = *reinterpret_cast<const u32x2*>(p.bias + cq);
    if (!epi_prefetches(p) || wrows != 64) return;      // (128 rows per wave: 128 accumulator registers leave no room for 64 more across the last slab)
    const float* r = reinterpret_cast<const float*>(p.res1) + cq;
    const int mlast = p.M - 1;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < npre) pre[k] = ld_stream(r + (long)min(rowq + 4 * k, mlast) * p.ldr1);     // (npre < 16: the rest is read in the epilogue)
}

// Row quads of the fp32 residual prefetched across the last slab: all 16, or 8 where eight waves share a CU's registers (256 per wave:
// without packed fp32 arithmetic the 256x128 tile's last slab had spilled three of them)
template <int NW>
constexpr int epi_npre() { return NW >= 8 ? 8 : 16; }

// LayerNorm fold (consumer side).  The GEMM ran on the RAW fp16 activations x with W' = gamma (.) W, so
//   LN(x) W^T + b = rstd (x W'^T - mean c) + d,   c_n = sum_k W'_nk,  d_n = sum_k beta_k W_nk + b_n
// (ln_mode 1: tokens are the rows of this GEMM; ln_mode 2 — the transposed V^T = W_v X^T product — tokens are its columns and
// c / d are per row).  The producer of x wrote, per token and 64-column block, (sum, M2 about the block mean): merged here with
// Chan's formula, so nothing cancels against the row mean.  Returns (mean * rstd, rstd) of one token.  The table holds an even
// number of blocks per token (whole 16-byte loads); these loads are issued after the mainloop (a folded GEMM replaces a whole
// LayerNorm launch: one exposed L2 round trip is a small price, and holding 40 more registers across the last slab is not).
__device__ __forceinline__ float2 ln_token_coeffs(const MmaParams& p, int token) {
    const int pairs = (p.ln_parts + 1) >> 1;
    const f32x4* s = reinterpret_cast<const f32x4*>(p.ln_stats) + (long)token * pairs;
    f32x4 t[LN_MAX_PARTS / 2];
#pragma unroll
    for (int q = 0; q < LN_MAX_PARTS / 2; ++q) t[q] = s[min(q, pairs - 1)];
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < LN_MAX_PARTS; ++q) tot += q < p.ln_parts ? ((q & 1) ? t[q >> 1].z : t[q >> 1].x) : 0.f;
    const float n = 64.f * (float)p.ln_parts, mean = tot / n;
    float m2 = 0.f;
#pragma unroll
    for (int q = 0; q < LN_MAX_PARTS; ++q) {
        const float sq = (q & 1) ? t[q >> 1].z : t[q >> 1].x, mq = (q & 1) ? t[q >> 1].w : t[q >> 1].y;
        const float dm = sq * (1.f / 64.f) - mean;
        m2 += q < p.ln_parts ? fmaf(64.f * dm, dm, mq) : 0.f;
    }
    const float r = rsqrtf(m2 / n + p.ln_eps);
    // The GEMM ran on the fp16 copy of x: its rounding, 2^-11 |x|, becomes (|mean| / sigma) 2^-11 of the normalised value.  Beyond
    // SG_LN_GUARD_RATIO the fold is no longer the LayerNorm the reference computes in fp32 — say so instead of returning it silently.
    if (p.ln_guard && fabsf(mean * r) > SG_LN_GUARD_RATIO) atomicOr(p.ln_guard, SG_LN_GUARD_OFFSET);
    return make_float2(mean * r, r);
}

union H4 { uint2 u; f16 h[4]; };


__device__ __forceinline__ void store_out4(const MmaParams& p, int gm, int gn, const float (&v)[4]) {
    const bool f32 = p.flags & SG_F_OUT_F32;
    if (f32) st_stream(reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn, f32x4{v[0], v[1], v[2], v[3]});
    if (!f32 || p.C2) {
        H4 o;
        // beside an fp32 output the fp16 copy is the operand of a folded LayerNorm (or a harvested feature): saturate it instead of
        // writing inf — the fp32 stream, which the reference's LayerNorm reads, is exact either way (the guard below reports it)
#pragma unroll
        for (int e = 0; e < 4; ++e) o.h[e] = (f16)(f32 ? __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f) : v[e]);
        if (!f32) st_stream(reinterpret_cast<f16*>(p.C) + (long)gm * p.ldc + gn, o.u);
        if (p.C2) st_stream(p.C2 + (long)gm * p.ldc2 + gn, o.u);
    }
}

// Everything after the last MFMA.  NW waves as a WGM x WGN grid, this wave at (wm, wn); m0 / n0 = the workgroup's tile origin.
// One workgroup barrier (the LDS ring is dead once every wave has left the last slab), then each wave works alone; a second
// barrier only when GroupNorm statistics are requested (to add up the WGM wave rows).
// WTM = 4 (round 6, mma_fat_kernel): a wave owns 128x64 = two 64x64 halves, finished one after the other through the SAME staging region
// (same-wave LDS traffic is in order); the residual of the second half is read here instead of prefetched, the GroupNorm partial sums of
// the halves are added in registers before the waves' rows meet in LDS (one partial per 128 WGM rows).
template <int WGM, int WGN, int WTM = 2>
__device__ __forceinline__ void epi_finish(const MmaParams& p, char* smem, f32x16 (&acc)[WTM][2], EPI_PRE_PARAMS,
                                           int m0, int n0, int z, int wave, int wm0, int wn, int lane) {
    constexpr int BN = 64 * WGN, NW = WGM * WGN, NT = 64 * NW, NH = WTM / 2;
    const int lnm = p.ln_mode;
    const int l31 = lane & 31, hi = lane >> 5, lr = lane >> 4, lc = lane & 15;
    float* stg = reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES);
    // the ring is dead once every wave has issued its last fragment reads (each wave's own reads were waited for by its MFMAs).
    // Raw s_barrier: __syncthreads() would also drain vmcnt, i.e. wait for the prefetched residual before the transpose starts.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float cs[4] = {0, 0, 0, 0}, cq2[4] = {0, 0, 0, 0};
    const bool want_stats = p.stats != nullptr;
#pragma unroll
    for (int half = 0; half < NH; ++half) {
    const int wm = wm0 * NH + half;              // this half's 64-row block of the tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(stg + (i * 32 + l31) * EPI_PITCH + j * 32 + 8 * g + 4 * hi) =
                    make_float4(acc[2 * half + i][j][4 * g], acc[2 * half + i][j][4 * g + 1], acc[2 * half + i][j][4 * g + 2],
                                acc[2 * half + i][j][4 * g + 3]);
    // same-wave LDS write -> read: in-order within the wave, the compiler's lgkmcnt wait covers it
    const int rowq = m0 + wm * 64 + lr, colq = n0 + wn * 64 + 4 * lc;
    float* xtr = stg + EPI_XTR;
    if (lnm) {               // lane l: coefficients of token row (mode 1) / token column (mode 2) l of this wave's sub-tile
        *reinterpret_cast<float2*>(xtr + 2 * lane) =
            ln_token_coeffs(p, lnm == 1 ? min(m0 + wm * 64 + lane, p.M - 1) : min(n0 + wn * 64 + lane, p.N - 1));
        if (lnm == 2) {      // ... and c / d of row l
            const int rr = min(m0 + wm * 64 + lane, p.M - 1);
            *reinterpret_cast<float2*>(xtr + 128 + 2 * lane) = make_float2(p.ln_c[rr], p.ln_d[rr]);
        }
    }
    if (p.mode == SG_EPI_GEGLU && p.splits == 1) {
        // interleaved layout (groups of 64 rows of W: 32 value rows then the 32 matching gate rows): the wave's columns 0..31 are
        // values, 32..63 the gates of the SAME 32 outputs.  Lane l: row 8 k + (l >> 3), values 4 (l & 7) .. +3.
        const int gr = lane >> 3, gc = (lane & 7) * 4;
        const int gv = n0 + wn * 64 + gc;                         // interleaved column of the 4 values; gates at gv + 32
        if (gv >= p.N) continue;
        float bv[4] = {0, 0, 0, 0}, bg[4] = {0, 0, 0, 0};
        if (p.bias) {
            H4 a, b;
            a.u = *reinterpret_cast<const uint2*>(p.bias + gv); b.u = *reinterpret_cast<const uint2*>(p.bias + gv + 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = (float)a.h[e]; bg[e] = (float)b.h[e]; }
        }
        const int go = (gv >> 6) * 32 + (gv & 31);                // interleaved column -> output column
        float cv[4] = {0, 0, 0, 0}, cg[4] = {0, 0, 0, 0};
        if (lnm == 1) {      // folded LayerNorm: value / gate = rstd (acc - mean c) + d  (d carries the bias)
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.ln_c + gv), b = *reinterpret_cast<const f32x4*>(p.ln_c + gv + 32);
            const f32x4 c = *reinterpret_cast<const f32x4*>(p.ln_d + gv), d = *reinterpret_cast<const f32x4*>(p.ln_d + gv + 32);
            cv[0] = a.x; cv[1] = a.y; cv[2] = a.z; cv[3] = a.w; cg[0] = b.x; cg[1] = b.y; cg[2] = b.z; cg[3] = b.w;
            bv[0] += c.x; bv[1] += c.y; bv[2] += c.z; bv[3] += c.w; bg[0] += d.x; bg[1] += d.y; bg[2] += d.z; bg[3] += d.w;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = 8 * k + gr, gm = m0 + wm * 64 + row;
            const float4 a = *reinterpret_cast<const float4*>(stg + row * EPI_PITCH + gc);
            const float4 b = *reinterpret_cast<const float4*>(stg + row * EPI_PITCH + 32 + gc);
            float2 mr = make_float2(0.f, 1.f);
            if (lnm == 1) mr = *reinterpret_cast<const float2*>(xtr + 2 * row);
            if (gm >= p.M) continue;
            const float va[4] = {a.x, a.y, a.z, a.w}, ga[4] = {b.x, b.y, b.z, b.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = (fmaf(mr.y, va[e], bv[e]) - mr.x * cv[e]) * gelu_erf_f(fmaf(mr.y, ga[e], bg[e]) - mr.x * cg[e]);
            store_out4(p, gm, go, o);
        }
        continue;
    }
    const bool col_ok = colq < p.N;
    if (p.splits > 1) {   // raw fp32 partial tile; the epilogue happens in splitk_reduce_kernel
        float* wsz = p.ws + (size_t)z * p.M * p.N + colq;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = 4 * k + lr, gm = rowq + 4 * k;
            const float4 v = *reinterpret_cast<const float4*>(stg + row * EPI_PITCH + 4 * lc);
            if (gm < p.M && col_ok) st_stream(wsz + (size_t)gm * p.N, f32x4{v.x, v.y, v.z, v.w});
        }
        continue;
    }
    // ---- fused linear epilogue: bias, temb row-bias, residuals (fp32 res1 prefetched), outputs, optional GroupNorm statistics.
    // Phases, not a per-row loop: all loads of a term are issued together (clamped addresses, no predicate), so the epilogue pays
    // one memory round trip per term instead of one per row quad.
    const bool r1f32 = p.flags & SG_F_RES1_F32, r2f32 = p.flags & SG_F_RES2_F32;
    // (a2 + h) + (a3 + h): the same tensor as both residuals is read once
    const bool res2_same = p.res2 != nullptr && p.res2 == p.res1 && p.ldr2 == p.ldr1 && r1f32 == r2f32;
    const int cq = min(colq, p.N - 4), mlast = p.M - 1;
    float v[16][4];
    {
        float bias4[4] = {0, 0, 0, 0};
        if (p.bias) {
            H4 b; b.u = make_uint2(pbias.x, pbias.y);
#pragma unroll
            for (int e = 0; e < 4; ++e) bias4[e] = (float)b.h[e];
        }
        if (lnm == 1) {           // rows are tokens: per-row (mean rstd, rstd), per-column c / d
            const f32x4 lc4 = *reinterpret_cast<const f32x4*>(p.ln_c + cq), ld4 = *reinterpret_cast<const f32x4*>(p.ln_d + cq);
            const float c4[4] = {lc4.x, lc4.y, lc4.z, lc4.w};
            const float d4[4] = {ld4.x + bias4[0], ld4.y + bias4[1], ld4.z + bias4[2], ld4.w + bias4[3]};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float4 a = *reinterpret_cast<const float4*>(stg + (4 * k + lr) * EPI_PITCH + 4 * lc);
                const float2 mr = *reinterpret_cast<const float2*>(xtr + 2 * (4 * k + lr));
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] = fmaf(mr.y, av[e], d4[e]) - mr.x * c4[e];
            }
        } else if (lnm == 2) {    // columns are tokens (V^T = W_v X^T): per-column (mean rstd, rstd), per-row c / d
            float mrx[4], mry[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float2 t = *reinterpret_cast<const float2*>(xtr + 2 * (4 * lc + e)); mrx[e] = t.x; mry[e] = t.y; }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float4 a = *reinterpret_cast<const float4*>(stg + (4 * k + lr) * EPI_PITCH + 4 * lc);
                const float2 cd = *reinterpret_cast<const float2*>(xtr + 128 + 2 * (4 * k + lr));
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] = fmaf(mry[e], av[e], cd.y + bias4[e]) - mrx[e] * cd.x;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float4 a = *reinterpret_cast<const float4*>(stg + (4 * k + lr) * EPI_PITCH + 4 * lc);
                v[k][0] = a.x + bias4[0]; v[k][1] = a.y + bias4[1]; v[k][2] = a.z + bias4[2]; v[k][3] = a.w + bias4[3];
            }
        }
    }
    // (loads of a term in batches of KB row quads: all 16 at once, or 8 + 8 where the register budget is 256 per wave)
    constexpr int KB = (NH > 1 || NW >= 8) ? 8 : 16;      // (eight waves per workgroup: 256 registers per wave)
    auto add_f32 = [&](const float* base, long ld) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < 16; kb += KB) {
            float4 t[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) t[k] = *reinterpret_cast<const float4*>(base + (long)min(rowq + 4 * (kb + k), mlast) * ld + cq);
#pragma unroll
            for (int k = 0; k < KB; ++k) { v[kb + k][0] += t[k].x; v[kb + k][1] += t[k].y; v[kb + k][2] += t[k].z; v[kb + k][3] += t[k].w; }
        }
    };
    auto add_f16 = [&](const f16* base, long ld) __attribute__((always_inline)) {
        uint2 t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = *reinterpret_cast<const uint2*>(base + (long)min(rowq + 4 * k, mlast) * ld + cq);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            H4 h; h.u = t[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] += (float)h.h[e];
        }
    };
    if (p.rowbias) {
#pragma unroll
        for (int kb = 0; kb < 16; kb += KB) {
            float4 t[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k)
                t[k] = *reinterpret_cast<const float4*>(p.rowbias + (long)fd_div((unsigned)min(rowq + 4 * (kb + k), mlast), p.fd_rpb) * p.rowbias_ld + cq);
#pragma unroll
            for (int k = 0; k < KB; ++k) { v[kb + k][0] += t[k].x; v[kb + k][1] += t[k].y; v[kb + k][2] += t[k].z; v[kb + k][3] += t[k].w; }
        }
    }
    if (p.res1) {
        if (r1f32) {
            const float w = res2_same ? 2.f : 1.f;       // x + r + r == x + 2 r up to one rounding of the fp32 stream
            if constexpr (NH > 1) {                       // 128 rows per wave: nothing was prefetched (registers); two batches of 8 row quads
                const float* r = reinterpret_cast<const float*>(p.res1) + cq;
#pragma unroll
                for (int kb = 0; kb < 16; kb += 8) {
                    f32x4 t[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = ld_stream(r + (long)min(rowq + 4 * (kb + k), mlast) * p.ldr1);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        v[kb + k][0] = fmaf(w, t[k].x, v[kb + k][0]); v[kb + k][1] = fmaf(w, t[k].y, v[kb + k][1]);
                        v[kb + k][2] = fmaf(w, t[k].z, v[kb + k][2]); v[kb + k][3] = fmaf(w, t[k].w, v[kb + k][3]);
                    }
                }
            } else {
                if constexpr (epi_npre<NW>() < 16) {
                    const float* r = reinterpret_cast<const float*>(p.res1) + cq;
#pragma unroll
                    for (int k = epi_npre<NW>(); k < 16; ++k) pre[k] = ld_stream(r + (long)min(rowq + 4 * k, mlast) * p.ldr1);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    v[k][0] = fmaf(w, pre[k].x, v[k][0]); v[k][1] = fmaf(w, pre[k].y, v[k][1]);
                    v[k][2] = fmaf(w, pre[k].z, v[k][2]); v[k][3] = fmaf(w, pre[k].w, v[k][3]);
                }
            }
        } else {
            add_f16(reinterpret_cast<const f16*>(p.res1), p.ldr1);
            if (res2_same) add_f16(reinterpret_cast<const f16*>(p.res1), p.ldr1);
        }
    }
    if (p.res2 && !res2_same) {
        if (r2f32) add_f32(reinterpret_cast<const float*>(p.res2), p.ldr2);
        else add_f16(reinterpret_cast<const f16*>(p.res2), p.ldr2);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int gm = rowq + 4 * k;
        if (gm < p.M && col_ok) {
            store_out4(p, gm, colq, v[k]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { cs[e] += v[k][e]; cq2[e] = fmaf(v[k][e], v[k][e], cq2[e]); }
        }
    }
    if (p.ln_out) {
        // LayerNorm fold, producer side: per token (row) and per 64-column block (this wave's columns) the sum and the M2 about the
        // block mean of the FINAL fp32 values.  The wave puts them back into its staging image and lane l reads row l whole
        // (16 x 16 B, conflict-free at this pitch): 64 adds + 64 fmas per lane, no cross-lane traffic, fixed order.
#pragma unroll
        for (int k = 0; k < 16; ++k)
            *reinterpret_cast<float4*>(stg + (4 * k + lr) * EPI_PITCH + 4 * lc) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
        float sum = 0.f, m2 = 0.f, mean = 0.f;
        if constexpr (NH > 1) {      // 128 rows per wave: the row is read twice, eight quads at a time (registers)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int qb = 0; qb < 16; qb += 8) {
                    float4 rv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) rv[q] = *reinterpret_cast<const float4*>(stg + lane * EPI_PITCH + 4 * (qb + q));
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (pass == 0) {
                            sum += (rv[q].x + rv[q].y) + (rv[q].z + rv[q].w);
                        } else {
                            const float a = rv[q].x - mean, b = rv[q].y - mean, c = rv[q].z - mean, d = rv[q].w - mean;
                            m2 = fmaf(a, a, m2); m2 = fmaf(b, b, m2); m2 = fmaf(c, c, m2); m2 = fmaf(d, d, m2);
                        }
                    }
                }
                if (pass == 0) mean = sum * (1.f / 64.f);
            }
        } else {
            float4 rv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) rv[q] = *reinterpret_cast<const float4*>(stg + lane * EPI_PITCH + 4 * q);
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += (rv[q].x + rv[q].y) + (rv[q].z + rv[q].w);
            mean = sum * (1.f / 64.f);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float a = rv[q].x - mean, b = rv[q].y - mean, c = rv[q].z - mean, d = rv[q].w - mean;
                m2 = fmaf(a, a, m2); m2 = fmaf(b, b, m2); m2 = fmaf(c, c, m2); m2 = fmaf(d, d, m2);
            }
        }
        const int gm = m0 + wm * 64 + lane;
        if (gm < p.M && n0 + wn * 64 < p.N) {
            *reinterpret_cast<float2*>(p.ln_out + ((size_t)gm * (((p.N >> 6) + 1) & ~1) + ((n0 >> 6) + wn)) * 2) = make_float2(sum, m2);
            // every |x| of the block is at most |mean| + sqrt(M2): below the fp16 maximum the raw copy did not saturate
            if (p.ln_guard && !(fabsf(mean) + sqrtf(m2) < 65504.f)) atomicOr(p.ln_guard, SG_LN_GUARD_RANGE);
        }
    }
    }       // halves
    if (want_stats && p.mode == SG_EPI_LINEAR && p.splits == 1) {
        float* xtr = stg + EPI_XTR;
        // GroupNorm statistics as an epilogue: per-(row tile, channel) sums of the FINAL fp32 values (bias / temb / residual
        // included, before the fp16 rounding).  A lane holds 16 rows of its 4 columns; the 4 lane groups (l >> 4) are added by two
        // exchanges, the WGM wave rows through LDS in fixed order: deterministic, no atomics.
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cs[e] += __shfl_xor(cs[e], 16, 64); cq2[e] += __shfl_xor(cq2[e], 16, 64);
            cs[e] += __shfl_xor(cs[e], 32, 64); cq2[e] += __shfl_xor(cq2[e], 32, 64);
        }
        // every wave parks its 64 columns x 2 planes in the second half of its own extra area
        if (lr == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xtr[128 + (4 * lc + e) * 2 + 0] = cs[e];
                xtr[128 + (4 * lc + e) * 2 + 1] = cq2[e];
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < BN * 2; idx += NT) {
            const int plane = idx / BN, col = idx - plane * BN;
            const int gn = n0 + col;
            if (gn < p.N) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WGM; ++w)
                    s += reinterpret_cast<const float*>(smem + (w * WGN + (col >> 6)) * EPI_WAVE_BYTES)[EPI_XTR + 128 + (col & 63) * 2 + plane];
                p.stats[((size_t)(m0 / (32 * WTM * WGM)) * 2 + plane) * p.N + gn] = s;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Epilogue of the 32x32-per-wave kernel (mma_lat_kernel, WT = 1 in mma_pipe_body).  The waves of a workgroup form 2x2 GROUPS; a group
// owns one 64x64 block of the tile: its four waves write their accumulators into ONE row-major fp32 image (same pitch as above), and
// after the barrier wave q of the group finishes rows 16 q .. 16 q + 15 of the block — the row-quad scheme of epi_finish with 4 quads
// per lane instead of 16 (instruction k: lane l holds columns 4 (l & 15) .. +3 of block row 16 q + 4 k + (l >> 4)), so every global
// access is still 4 rows x 256 contiguous bytes.  Everything the linear epilogue of epi_finish offers except GEGLU: bias, temb row
// bias, two residuals, fp32 / fp16 outputs, the LayerNorm fold on both sides, GroupNorm partials (one per tile: 32 WGM rows), split-K
// partial tiles.  Row sums for the LayerNorm partials come from 16-lane exchanges (a row of the strip lives in 16 lanes).
template <int WGN>
__device__ __forceinline__ void epi_prefetch_q(const MmaParams& p, int m0, int n0, int wave, int lane, f32x4 (&pre)[4], u32x2& pbias) {
    if (p.splits > 1 || p.mode != SG_EPI_LINEAR) return;
    const int wm32 = wave / WGN, wn32 = wave % WGN, q = (wm32 & 1) * 2 + (wn32 & 1);
    const int rowq = m0 + (wm32 >> 1) * 64 + q * 16 + (lane >> 4), cq = min(n0 + (wn32 >> 1) * 64 + 4 * (lane & 15), p.N - 4);
    if (p.bias) pbias = *reinterpret_cast<const u32x2*>(p.bias + cq);
    if (!epi_prefetches(p)) return;
    const float* r = reinterpret_cast<const float*>(p.res1) + cq;
    const int mlast = p.M - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) pre[k] = ld_stream(r + (long)min(rowq + 4 * k, mlast) * p.ldr1);
}

template <int WGM, int WGN>
__device__ __forceinline__ void epi_finish_q(const MmaParams& p, char* smem, f32x16& acc, f32x4 (&pre)[4], u32x2& pbias,
                                             int m0, int n0, int z, int wave, int lane) {
    constexpr int GN_ = WGN / 2, NW = WGM * WGN, BM = 32 * WGM, BN = 32 * WGN, NT = 64 * NW;
    const int wm32 = wave / WGN, wn32 = wave % WGN;
    const int gm_ = wm32 >> 1, gn_ = wn32 >> 1, wi = wm32 & 1, wj = wn32 & 1, q = wi * 2 + wj;
    const int l31 = lane & 31, hi = lane >> 5, lr = lane >> 4, lc = lane & 15;
    const int lnm = p.ln_mode;
    char* grp = smem + (gm_ * GN_ + gn_) * EPIQ_GROUP_BYTES;
    float* stg = reinterpret_cast<float*>(grp);
    float* xtr = reinterpret_cast<float*>(grp + 64 * EPI_PITCH * 4 + q * 1024);     // this wave's own 1 KB
    const int rbase = m0 + gm_ * 64 + q * 16, cbase = n0 + gn_ * 64;                 // first row of the strip, first column of the block
    // the ring is dead once every wave has issued its last fragment reads (raw barrier: a prefetched residual stays in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(stg + (wi * 32 + l31) * EPI_PITCH + wj * 32 + 8 * g + 4 * hi) =
            make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    // LayerNorm coefficients depend on no other wave: computed while the other waves are still writing the image
    if (lnm == 1) {          // rows are tokens: lane l -> strip row l & 15 (the four lane rows compute the same value; one stores)
        const float2 c = ln_token_coeffs(p, min(rbase + lc, p.M - 1));
        if (lr == 0) *reinterpret_cast<float2*>(xtr + 2 * lc) = c;
    } else if (lnm == 2) {   // columns are tokens: lane l -> block column l; c / d of the strip's rows
        *reinterpret_cast<float2*>(xtr + 2 * lane) = ln_token_coeffs(p, min(cbase + lane, p.N - 1));
        const int rr = min(rbase + lc, p.M - 1);
        if (lr == 0) *reinterpret_cast<float2*>(xtr + 128 + 2 * lc) = make_float2(p.ln_c[rr], p.ln_d[rr]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int rowq = rbase + lr, colq = cbase + 4 * lc;
    const bool col_ok = colq < p.N;
    const float* img = stg + (q * 16 + lr) * EPI_PITCH + 4 * lc;             // + 4 k rows
    if (p.splits > 1) {      // raw fp32 partial tile; the epilogue happens in splitk_reduce_kernel (or in the consumer)
        float* wsz = p.ws + (size_t)z * p.M * p.N + colq;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gm = rowq + 4 * k;
            const float4 v = *reinterpret_cast<const float4*>(img + 4 * k * EPI_PITCH);
            if (gm < p.M && col_ok) st_stream(wsz + (size_t)gm * p.N, f32x4{v.x, v.y, v.z, v.w});
        }
        return;
    }
    const bool want_stats = p.stats != nullptr;
    const bool r1f32 = p.flags & SG_F_RES1_F32, r2f32 = p.flags & SG_F_RES2_F32;
    const bool res2_same = p.res2 != nullptr && p.res2 == p.res1 && p.ldr2 == p.ldr1 && r1f32 == r2f32;
    const int cq = min(colq, p.N - 4), mlast = p.M - 1;
    float v[4][4];
    {
        float bias4[4] = {0, 0, 0, 0};
        if (p.bias) {
            H4 b; b.u = make_uint2(pbias.x, pbias.y);
#pragma unroll
            for (int e = 0; e < 4; ++e) bias4[e] = (float)b.h[e];
        }
        if (lnm == 1) {
            const f32x4 lc4 = *reinterpret_cast<const f32x4*>(p.ln_c + cq), ld4 = *reinterpret_cast<const f32x4*>(p.ln_d + cq);
            const float c4[4] = {lc4.x, lc4.y, lc4.z, lc4.w};
            const float d4[4] = {ld4.x + bias4[0], ld4.y + bias4[1], ld4.z + bias4[2], ld4.w + bias4[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 a = *reinterpret_cast<const float4*>(img + 4 * k * EPI_PITCH);
                const float2 mr = *reinterpret_cast<const float2*>(xtr + 2 * (4 * k + lr));
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] = fmaf(mr.y, av[e], d4[e]) - mr.x * c4[e];
            }
        } else if (lnm == 2) {
            float mrx[4], mry[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float2 t = *reinterpret_cast<const float2*>(xtr + 2 * (4 * lc + e)); mrx[e] = t.x; mry[e] = t.y; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 a = *reinterpret_cast<const float4*>(img + 4 * k * EPI_PITCH);
                const float2 cd = *reinterpret_cast<const float2*>(xtr + 128 + 2 * (4 * k + lr));
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] = fmaf(mry[e], av[e], cd.y + bias4[e]) - mrx[e] * cd.x;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 a = *reinterpret_cast<const float4*>(img + 4 * k * EPI_PITCH);
                v[k][0] = a.x + bias4[0]; v[k][1] = a.y + bias4[1]; v[k][2] = a.z + bias4[2]; v[k][3] = a.w + bias4[3];
            }
        }
    }
    auto add_f32 = [&](const float* base, long ld) __attribute__((always_inline)) {
        float4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const float4*>(base + (long)min(rowq + 4 * k, mlast) * ld + cq);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k][0] += t[k].x; v[k][1] += t[k].y; v[k][2] += t[k].z; v[k][3] += t[k].w; }
    };
    auto add_f16 = [&](const f16* base, long ld) __attribute__((always_inline)) {
        uint2 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const uint2*>(base + (long)min(rowq + 4 * k, mlast) * ld + cq);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            H4 h; h.u = t[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] += (float)h.h[e];
        }
    };
    if (p.rowbias) {
        float4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            t[k] = *reinterpret_cast<const float4*>(p.rowbias + (long)fd_div((unsigned)min(rowq + 4 * k, mlast), p.fd_rpb) * p.rowbias_ld + cq);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k][0] += t[k].x; v[k][1] += t[k].y; v[k][2] += t[k].z; v[k][3] += t[k].w; }
    }
    if (p.res1) {
        if (r1f32) {
            const float w = res2_same ? 2.f : 1.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k][0] = fmaf(w, pre[k].x, v[k][0]); v[k][1] = fmaf(w, pre[k].y, v[k][1]);
                v[k][2] = fmaf(w, pre[k].z, v[k][2]); v[k][3] = fmaf(w, pre[k].w, v[k][3]);
            }
        } else {
            add_f16(reinterpret_cast<const f16*>(p.res1), p.ldr1);
            if (res2_same) add_f16(reinterpret_cast<const f16*>(p.res1), p.ldr1);
        }
    }
    if (p.res2 && !res2_same) {
        if (r2f32) add_f32(reinterpret_cast<const float*>(p.res2), p.ldr2);
        else add_f16(reinterpret_cast<const f16*>(p.res2), p.ldr2);
    }
    float cs[4] = {0, 0, 0, 0}, cq2[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gm = rowq + 4 * k;
        if (gm < p.M && col_ok) {
            store_out4(p, gm, colq, v[k]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { cs[e] += v[k][e]; cq2[e] = fmaf(v[k][e], v[k][e], cq2[e]); }
        }
    }
    if (p.ln_out) {
        // LayerNorm fold, producer side: (sum, M2 about the block mean) of the FINAL fp32 values per token and 64-column block.  A row of
        // the strip lives in the 16 lanes that share lane >> 4: four exchanges per quantity, fixed order.
        float s4[4], m4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s4[k] = (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int k = 0; k < 4; ++k) s4[k] += __shfl_xor(s4[k], o, 64);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float mean = s4[k] * (1.f / 64.f);
            const float a = v[k][0] - mean, b = v[k][1] - mean, c = v[k][2] - mean, d = v[k][3] - mean;
            m4[k] = fmaf(a, a, fmaf(b, b, fmaf(c, c, d * d)));
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int k = 0; k < 4; ++k) m4[k] += __shfl_xor(m4[k], o, 64);
        if (lc == 0 && cbase < p.N) {
            const int nblk = ((p.N >> 6) + 1) & ~1, blk = (n0 >> 6) + gn_;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int gm = rowq + 4 * k;
                if (gm < p.M) {
                    *reinterpret_cast<float2*>(p.ln_out + ((size_t)gm * nblk + blk) * 2) = make_float2(s4[k], m4[k]);
                    if (p.ln_guard && !(fabsf(s4[k] * (1.f / 64.f)) + sqrtf(m4[k]) < 65504.f)) atomicOr(p.ln_guard, SG_LN_GUARD_RANGE);
                }
            }
        }
    }
    if (want_stats) {
        // GroupNorm statistics: per-(tile, channel) sums of the final fp32 values.  A lane holds 4 rows of its 4 columns; the 4 lane rows
        // are added by two exchanges, the strips of the tile (4 per group, WGM / 2 groups per column block) through LDS in fixed order.
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cs[e] += __shfl_xor(cs[e], 16, 64); cq2[e] += __shfl_xor(cq2[e], 16, 64);
            cs[e] += __shfl_xor(cs[e], 32, 64); cq2[e] += __shfl_xor(cq2[e], 32, 64);
        }
        if (lr == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xtr[128 + (4 * lc + e) * 2 + 0] = cs[e];
                xtr[128 + (4 * lc + e) * 2 + 1] = cq2[e];
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < BN * 2; idx += NT) {
            const int plane = idx / BN, col = idx - plane * BN;
            const int gn = n0 + col;
            if (gn < p.N) {
                float s = 0.f;
#pragma unroll
                for (int g = 0; g < WGM / 2; ++g)
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        s += reinterpret_cast<const float*>(smem + (g * GN_ + (col >> 6)) * EPIQ_GROUP_BYTES + 64 * EPI_PITCH * 4 + w * 1024)[128 + (col & 63) * 2 + plane];
                p.stats[((size_t)(m0 / BM) * 2 + plane) * p.N + gn] = s;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Generic kernel: 256 threads (2x2 waves), register-staged double buffer, zero-fill predicates.
template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256) void mma_kernel(const MmaParams p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    static_assert(TM == 2 && TN == 2, "epi_finish works on 64x64 wave tiles");
    constexpr int A_IT = BM / 32, B_IT = BN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int EPI_BYTES = 4 * EPI_WAVE_BYTES;   // staging regions (+ 1 KB of scratch each) of epi_finish
    constexpr int SMEM = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    int m0, n0, z;
    decode_block(p, BM, BN, m0, n0, z);
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);   // transposed product: see epi_finish
        }
        if (more) store_lds(buf ^ 1);
        __syncthreads();
    }
    EPI_PRE_DECL;
    epi_prefetch(p, m0, n0, wm, wn, lane, EPI_PRE_ARGS);
    __syncthreads();
    epi_finish<2, 2>(p, smem, acc, EPI_PRE_ARGS, m0, n0, z, wave, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------------------
// Pipelined kernel: WGM x WGN waves of 64x64, 3 LDS stages filled by LDS-DMA.  Requires loads that need no
// predicate: rows beyond M / N are clamped to the last valid row (their results are discarded by the epilogue);
// K % 64 == 0; for the convolution the input has a one-pixel zero border ([B, H+2, W+2, C]), so every tap of every
// output pixel reads valid memory and padding costs nothing.
// Slab t+2 is issued during slab t (behind its first k-step, or spread over its k-steps: SPREAD / STAGGER in mma_pipe_body), i.e. after the
// barrier that (a) publishes slab t (every wave waited for its own DMA with a counted vmcnt first) and (b) retires every wave's reads of
// slab t-1, whose stage it overwrites.
// One LDS-DMA piece (1 KiB per wave): 16 bytes per lane from `base` (wave-uniform) + `offb` (per-lane BYTE offset, < 2^32: the host
// validates element offsets < 2^31) to the wave-uniform LDS address `lds_wave_base` (+ lane * 16, the hardware's lane-linear image).
// SG_GLDS_SADDR (A/B build): the scalar-base form of the instruction, written out — the builtin always materialises a 64-bit per-lane
// address (one v_lshl_add_u64 per piece beside the MFMAs).  Every piece of a kernel must then go through here (M0 is set by hand).
struct LdsRef { char* p; unsigned a; };          // one LDS location as a generic pointer and as its LDS byte address
__device__ __forceinline__ LdsRef operator+(LdsRef r, int d) { return LdsRef{r.p + d, r.a + (unsigned)d}; }
__device__ __forceinline__ LdsRef lds_ref(char* smem) {
    return LdsRef{smem, (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long)((__attribute__((address_space(3))) char*)smem))};
}
__device__ __forceinline__ void glds16(const f16* base, unsigned offb, LdsRef dst) {
#ifdef SG_GLDS_SADDR
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(offb), "s"(base), "s"(dst.a) : "memory");
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + offb),
                                     (__attribute__((address_space(3))) void*)dst.p, 16, 0, 0);
#endif
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS ring depth S: 3 by default (one slab computing, two in flight).  S = 2 (round 4) halves the bytes in flight but brings the
// 128x128 and 256x64 tiles down to 74 / 80 KB of LDS, so that TWO workgroups — one of each of the step's two concurrent passes — fit a
// CU (round 2 measured the forced 128x128 / 2-stage configuration as the fastest whole step; development option pipe_stages).
template <int WGM, int WGN, int S, int WT = 2, int WTM = WT>
constexpr int pipe_smem_bytes() {
    constexpr int ring = S * (32 * WTM * WGM + 32 * WT * WGN) * 128;
    constexpr int epi = WT == 2 ? WGM * WGN * EPI_WAVE_BYTES : (WGM / 2) * (WGN / 2) * EPIQ_GROUP_BYTES;
    return ring > epi ? ring : epi;
}

// The slabs of one trip through the ring, each with its stage as a compile-time constant (mma_pipe_body): stage ST handles slab it + ST
// while that slab is a steady one.
template <int ST, int S, int YS, class F>
__device__ __forceinline__ void slab_seq(F& slab, int it, int nsteady) {
    if constexpr (ST < S) {
        if (it + ST < nsteady) {
            slab(it + ST, std::false_type{}, std::integral_constant<int, YS>{}, std::integral_constant<int, ST>{});
            slab_seq<ST + 1, S, YS>(slab, it, nsteady);
        }
    }
}
// The tail: slab nt - 1 - Y has Y younger slabs in flight (Y = TAIL - 1 ... 0; the last one, Y = 0, carries the epilogue's prefetch)
template <int Y, int S, class F>
__device__ __forceinline__ void slab_tail(F& slab, int nt) {
    if (nt > Y) slab(nt - 1 - Y, std::integral_constant<bool, Y == 0>{}, std::integral_constant<int, Y>{}, (nt - 1 - Y) % S);
    if constexpr (Y > 0) slab_tail<Y - 1, S>(slab, nt);
}

template <int WGM, int WGN, bool CONV, bool PROF = false, int S = 3, int WT = 2, int WTMX = WT>
__device__ __forceinline__ void mma_pipe_body(const MmaParams& p, char* smem) {
    // PROF (SG_BUILD_EXPERIMENTS): s_memtime stamps around the phases of every slab, summed per wave (sg_debug_gemm_anatomy /
    // _conv_anatomy): [0] slabs [1] vmcnt wait [2] barrier [3] first fragment reads + k-step 0 [4] k-step 1 up to the DMA issue
    // [5] DMA issue [6] rest of the slab [7] prologue (entry -> loop) [8] epilogue (loop end -> exit)
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
    // WT = 2: every wave owns 64x64 of the tile (2x2 MFMA accumulators; the throughput shapes).  WT = 1 (round 6, mma_lat_kernel): 32x32 per
    // wave — a 64x64 tile is shared by FOUR waves, so a slab costs each of them 4 MFMAs and 4 LDS-DMA pieces instead of 16 and 16, and the
    // ring is 4 - 8 stages deep: the form for the launches whose time is their dependent chain of slabs, not their FLOPs (below).
    // WTMX = 4 with WT = 2 (round 6, mma_fat_kernel): 128x64 per wave — 6 fragment reads and 0.25 LDS-DMA pieces per MFMA instead of 8 and 0.375
    constexpr int WTM = WTMX, WTN = WT;
    static_assert(WTM == WT || (WT == 2 && WTM == 4), "wave tiles: 32x32, 64x64 or 128x64");
    constexpr int NW = WGM * WGN, WM = 32 * WTM, WN = 32 * WTN, BM = WM * WGM, BN = WN * WGN;
    constexpr int A_IT = BM / (8 * NW), B_IT = BN / (8 * NW), LPT = A_IT + B_IT;   // LDS-DMA instructions / lane / slab
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int ISTR = NW * 1024;   // LDS bytes covered by one DMA instruction of the whole workgroup (8 rows / wave)
    // Refill placement (round 4, profiles/r04p_*): with ONE wave per SIMD (tiles of <= 4 waves) nothing covers the 55 - 90 cycles a wave
    // spends issuing each LDS-DMA piece, and a burst of 4 - 6 of them behind the first k-step drains the matrix pipe; one or two pieces
    // behind every k-step measured -2 ... -12 % per launch on the 4-wave tiles (GEMM 12288x320x1280 20.1 -> 17.6 us, conv 64^2 320->320 on
    // 256x64 40.6 -> 37.4).  The 8-wave 256x128 tile (two waves per SIMD cover each other) measured +3 ... +6 % and keeps the burst.
    // There the two waves of a SIMD leave the barrier together and would burst together: waves 4 - 7 issue theirs behind the THIRD k-step
    // instead (STAGGER: conv 64^2 640->320 92.8 -> 85.5 us, 64^2 320->320 49.3 -> 47.4, 16^2 / 32^2 -2 %, profiles/r04r_*).
#ifdef SG_PIPE_BURST
    constexpr bool SPREAD = false, STAGGER = false;       // A/B build (tools/ab_lib.py): rounds 1-4, every wave bursts behind the first k-step
#else
    constexpr bool SPREAD = (NW <= 4 || WT == 1) && !PROF, STAGGER = NW == 8 && WT == 2 && !PROF;
#endif
    static_assert(S >= 2 && S <= 8, "ring depth");
    static_assert(WT == 2 || (WGM % 2 == 0 && WGN % 2 == 0), "32x32 waves come in 2x2 groups (epi_finish_q)");
    static_assert((S - 1) * LPT < 64, "vmcnt is a 6-bit counter");
    static_assert(S * STAGE <= 160 * 1024, "the ring must fit the 160 KB of LDS");
    static_assert(S * STAGE <= pipe_smem_bytes<WGM, WGN, S, WT, WTM>(), "the LDS block covers the ring and the epilogue's staging regions");

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, hi = lane >> 5;
    int m0, n0, z;
    decode_block(p, BM, BN, m0, n0, z);
    const int kt0 = z * p.kt_per_split;
    const int nt = min(p.KT, kt0 + p.kt_per_split) - kt0;

    // staging coordinates of this lane for DMA instruction i: tile row srow + 8*NW*i, LDS slot (lane & 7).
    // All per-lane address arithmetic is done ONCE here as 32-bit element offsets; per slab only wave-uniform (scalar)
    // terms change: GEMM  A + kt*64;  conv  A + ((ky*wp + kx)*lda + cc*64)  — or, with nearest-2x upsampling, where the
    // source row (oy-1+ky)>>1 is not affine in ky, one of three precomputed row / column offsets picked by (ky, kx).
    const int srow = wave * 8 + (lane >> 3);
    const int wp = p.Wd + 2;
    // the weight side first: its offsets need no division, so the first DMA instructions leave before the pixel arithmetic
    unsigned w_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = srow + 8 * NW * i;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        w_off[i] = 2u * (unsigned)((long)min(n0 + row, p.N - 1) * p.ldw + lc * 8);       // bytes
    }
    // `sel` < 0: every piece; otherwise only the pieces whose running index (A pieces first, then W) is sel modulo 4 — the refill of a
    // slab spread over its four k-steps (SG_PIPE_SPREAD builds)
    // K order of the CONVOLUTION (round 5): slab kt = (channel block kt / 9, tap kt % 9) — the nine taps of one 64-channel block are
    // consecutive slabs.  They re-read the same input pixels (shifted by one): with the taps innermost that working set is one channel
    // block of the tile's pixels per workgroup (~40 KB; ~1.2 MB for the workgroups of an XCD) and stays in the XCD's 4 MB L2, where the
    // former order (tap kt / cpt outermost, all channel blocks inside) walked the whole input panel between two taps and re-fetched it
    // from the fabric nine times (profiles/traffic.json of the round-5 mid build: 1 053 MB per launch against 79 MB algorithmic for
    // the batch-20 16x16 convolutions).  The weights are [Cout][ky][kx][Cin]: slab kt starts at element (kt % 9) * Cin + (kt / 9) * 64
    // of a row — either order is a plain offset, nothing is repacked.  -DSG_CONV_TAP_MAJOR rebuilds the former order (A/B builds).
    // GEMM: slab kt starts at element kt * 64 of both operands.
    auto w_koff = [&](int kt) __attribute__((always_inline)) -> long {
        if constexpr (CONV) {
#ifdef SG_CONV_TAP_MAJOR
            return (long)kt * BK;
#else
            const int cc = (kt * 7282) >> 16, tap = kt - 9 * cc;        // kt / 9 for kt < 3 000 (validated on the host)
            return (long)tap * p.cpt * BK + (long)cc * BK;
#endif
        } else {
            return (long)kt * BK;
        }
    };
    const LdsRef lds = lds_ref(smem);
    auto issue_w = [&](long koff, int stage, int sel = -1) __attribute__((always_inline)) {
        const LdsRef sB = lds + (stage * STAGE + A_BYTES + wave * 1024);
        const f16* Wt = p.W + koff;
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (sel < 0 || ((A_IT + i) & 3) == sel) glds16(Wt, w_off[i], sB + i * ISTR);
    };
    if (nt > 0) issue_w(w_koff(kt0), 0);

    unsigned a_off[A_IT];                       // BYTE offset of the row (GEMM) / of tap (0, 0) (conv)
    unsigned a_par[A_IT];                       // conv with upsampling: parity bits of (oy - 1, ox - 1)
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int row = srow + 8 * NW * i;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);          // logical chunk this lane fetches
        const int gm = min(m0 + row, p.M - 1);
        if constexpr (CONV) {
            const int hw = p.Ho * p.Wo;
            const int b = (int)fd_div((unsigned)gm, p.fd_hw), rem = gm - b * hw;
            const int oy = (int)fd_div((unsigned)rem, p.fd_wo), ox = rem - oy * p.Wo;
            const long img = (long)b * (p.H + 2) * wp;
            // padded input pixel of tap (ky, kx): row ((oy*stride - 1 + ky) >> ups) + 1, column likewise
            if (!p.ups) {
                a_off[i] = 2u * (unsigned)((img + (long)(oy * p.stride) * wp + ox * p.stride) * p.lda + lc * 8);
                a_par[i] = 0;
            } else {
                // source row of tap ky: ((oy - 1 + ky) >> 1) + 1 = ((oy - 1) >> 1) + 1 + ((ky + ((oy - 1) & 1)) >> 1)
                a_off[i] = 2u * (unsigned)((img + (long)(((oy - 1) >> 1) + 1) * wp + ((ox - 1) >> 1) + 1) * p.lda + lc * 8);
                a_par[i] = (unsigned)(((oy - 1) & 1) | (((ox - 1) & 1) << 1));
            }
        } else {
            a_off[i] = 2u * (unsigned)((long)gm * p.lda + lc * 8);
            a_par[i] = 0;
        }
    }

    // source of slab kt's A operand: the wave-uniform base pointer (and, with nearest-2x upsampling, the tap), computed ONCE per slab;
    // a_emit then issues the pieces `sel` selects (sel < 0: all; otherwise running piece index == sel modulo 4)
    struct ABase { const f16* At; int ky, kx; long wk; };      // wk = element offset of the slab's weights in a row of W (w_koff)
    auto a_base = [&](int kt) __attribute__((always_inline)) {
        ABase r;
        r.ky = r.kx = 0;
        r.wk = w_koff(kt);
        if constexpr (CONV) {
#ifdef SG_CONV_TAP_MAJOR
            const int tap = (int)fd_div((unsigned)kt, p.fd_cpt), cc = kt - tap * p.cpt;
#else
            const int cc = (kt * 7282) >> 16, tap = kt - 9 * cc;
#endif
            const int ky = (tap * 11) >> 5, kx = tap - ky * 3;           // tap / 3 for tap < 9
            r.ky = ky; r.kx = kx;
            r.At = !p.ups ? p.A + ((long)(ky * wp + kx) * p.lda + cc * BK) : p.A + cc * BK;
        } else {
            r.At = p.A + kt * BK;
        }
        return r;
    };
    auto a_emit = [&](const ABase& ab, int stage, int sel) __attribute__((always_inline)) {
        const LdsRef sA = lds + (stage * STAGE + wave * 1024);
        if constexpr (CONV) {
            if (!p.ups) {
#pragma unroll
                for (int i = 0; i < A_IT; ++i)
                    if (sel < 0 || (i & 3) == sel) glds16(ab.At, a_off[i], sA + i * ISTR);
            } else {
                const unsigned rs = (unsigned)(wp * (int)p.lda), cs = (unsigned)p.lda;     // < 2^24 (validated on the host)
#pragma unroll
                for (int i = 0; i < A_IT; ++i) {
                    if (!(sel < 0 || (i & 3) == sel)) continue;
                    const unsigned dy = ((unsigned)ab.ky + (a_par[i] & 1u)) >> 1, dx = ((unsigned)ab.kx + (a_par[i] >> 1)) >> 1;
                    glds16(ab.At, a_off[i] + 2u * (__umul24(dy, rs) + __umul24(dx, cs)), sA + i * ISTR);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                if (sel < 0 || (i & 3) == sel) glds16(ab.At, a_off[i], sA + i * ISTR);
        }
    };
#ifdef SG_PIPE_BASE_DIV        // A/B build (tools/ab_lib.py): every slab's base from its index (multiply-shift division + 64-bit products)
    auto a_next = [&](int kt) __attribute__((always_inline)) { return a_base(kt); };
#else
    // The slabs of a block are requested in order (kt0, kt0 + 1, ...): their operand bases come from running counters — channel block,
    // tap column, tap row, element offset — instead of a division and a 64-bit product per slab (≈ 45 -> ≈ 10 scalar instructions in
    // front of every refill; the refill of the 256x128 tile is a burst right behind them)
    long it_off, it_wk;
#ifdef SG_CONV_TAP_MAJOR
    int it_cc = 0;
#endif
    int it_kx = 0, it_ky = 0;
    {
        const ABase b0 = a_base(kt0);
        it_off = b0.At - p.A;
        it_wk = b0.wk;
        if constexpr (CONV) {
#ifdef SG_CONV_TAP_MAJOR
            const int tap = (int)fd_div((unsigned)kt0, p.fd_cpt);
            it_cc = kt0 - tap * p.cpt;
#endif
            it_ky = b0.ky; it_kx = b0.kx;
        }
    }
#ifdef SG_CONV_TAP_MAJOR
    const long it_dx = CONV ? (p.ups ? 0 : (long)p.lda) - (long)p.cpt * BK : 0;       // next tap column: one pixel right, channel block 0
    const long it_dy = CONV && !p.ups ? (long)(wp - 3) * p.lda : 0;                    // ... next tap row: from column 3 back to 0, one row down
    auto a_next = [&](int) __attribute__((always_inline)) {
        ABase r;
        r.At = p.A + it_off; r.ky = it_ky; r.kx = it_kx; r.wk = it_wk;
        it_off += BK; it_wk += BK;
        if constexpr (CONV) {
            if (++it_cc == p.cpt) {
                it_cc = 0;
                it_off += it_dx;
                if (++it_kx == 3) { it_kx = 0; ++it_ky; it_off += it_dy; }
            }
        }
        return r;
    };
#else
    // taps innermost: one pixel right per slab; after column 2 one row down and back to column 0; after tap 8 back to tap 0 of the
    // next channel block.  (Nearest-2x upsampling: the pixel offset comes from (ky, kx) in a_emit, only the channel block moves here.)
    const long it_px = CONV && !p.ups ? (long)p.lda : 0;
    const long it_dy = CONV && !p.ups ? (long)(wp - 3) * p.lda : 0;
    const long it_dc = CONV ? (p.ups ? 0 : -3L * wp * p.lda) + BK : 0;
    const long it_wtap = CONV ? (long)p.cpt * BK : BK, it_wdc = CONV ? BK - 9L * p.cpt * BK : 0;
    auto a_next = [&](int) __attribute__((always_inline)) {
        ABase r;
        r.At = p.A + it_off; r.ky = it_ky; r.kx = it_kx; r.wk = it_wk;
        it_wk += it_wtap;
        if constexpr (CONV) {
            it_off += it_px;
            if (++it_kx == 3) {
                it_kx = 0; it_off += it_dy;
                if (++it_ky == 3) { it_ky = 0; it_off += it_dc; it_wk += it_wdc; }
            }
        } else {
            it_off += BK;
        }
        return r;
    };
#endif
#endif
    // (A pieces of a slab, then its W pieces: one a_next per slab, in slab order)
    auto issue_aw = [&](int kt, int stage, bool with_w) __attribute__((always_inline)) {
        const ABase ab = a_next(kt);
        a_emit(ab, stage, -1);
        if (with_w) issue_w(ab.wk, stage);
    };
    if (nt > 0) issue_aw(kt0, 0, false);                   // (its weights left first, above)
#pragma unroll
    for (int st = 1; st < S - 1; ++st)
        if (nt > st) issue_aw(kt0 + st, st, true);

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 pre_f[WT == 2 ? 16 : 4]; u32x2 pre_bias;       // prefetched epilogue operands (epi_prefetch / epi_prefetch_q)
    auto prefetch = [&]() __attribute__((always_inline)) {
        if constexpr (WT == 2) epi_prefetch(p, m0, n0, wm, wn, lane, pre_f, pre_bias, WM, epi_npre<NW>());
        else epi_prefetch_q<WGN>(p, m0, n0, wave, lane, pre_f, pre_bias);
    };

    // LDS offsets of this lane's fragment reads, per ring stage and k-step (the XOR swizzle makes the k-step term per-lane, and stages
    // 1 / 2 of the larger tiles lie beyond the 16-bit immediate of ds_read): 2 x 4 x S registers, opaque to the compiler so that it keeps
    // them instead of re-deriving base + constant with a v_add / v_or before every read of every slab
    // (WT = 1, deep rings of small stages: SPB consecutive stages share one base register — their distance fits ds_read's 16-bit immediate)
    constexpr int SPB = WT == 2 ? 1 : ((65536 - 8192) / STAGE > 0 ? (65536 - 8192) / STAGE : 1);
    constexpr int NBASE = (S + SPB - 1) / SPB;
    unsigned fa_off[NBASE][4], fb_off[NBASE][4];
#pragma unroll
    for (int st = 0; st < NBASE; ++st)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fa_off[st][ks] = (unsigned)(st * SPB * STAGE + lds_off(wm * WM + l31, ks * 2 + hi));
            fb_off[st][ks] = (unsigned)(st * SPB * STAGE + A_BYTES + lds_off(wn * WN + l31, ks * 2 + hi));
            asm volatile("" : "+v"(fa_off[st][ks]), "+v"(fb_off[st][ks]));
        }
    stamp(7);
    // one K slab; LAST = the final slab of this block (no refill: the residual of the epilogue is requested under its MFMAs instead);
    // YOUNG = the number of younger slabs in flight while this one is waited for (S - 2 in the steady loop, fewer at the tail).
    // `stage_c` = the ring stage of slab `it`: a std::integral_constant in the steady loop (round 4: S copies of the slab per trip, so the
    // 16 fragment reads address LDS as loop-invariant per-lane bases + immediates and the DMA destinations are constants — a run-time
    // stage cost 18 - 26 v_add_u32 per slab and wave, issued BESIDE the MFMAs, where VALU time adds to matrix time,
    // tools/probes/README.md), a plain int for the tail slabs.
    auto slab = [&](int it, auto last_tag, auto young_tag, auto stage_c) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int YOUNG = decltype(young_tag)::value;
        const int stage = stage_c;
        wait_vmcnt<YOUNG * LPT>();
        stamp(1);
        __builtin_amdgcn_s_barrier();
        stamp(2);
        const char* sA = smem + stage * STAGE;
        const char* sB = sA + A_BYTES;
        // fragment reads run one k-step ahead of the MFMAs that consume them (two register sets), so that only the first
        // read of a slab exposes LDS latency; the other three hide behind the previous k-step's four MFMAs
        f16x8 af[2][WTM], bf[2][WTN];
        auto load_frags = [&](int buf, int ks) __attribute__((always_inline)) {
            if constexpr (std::is_same<decltype(stage_c), int>::value) {       // run-time stage (tail slabs): addresses computed here
#pragma unroll
                for (int i = 0; i < WTM; ++i)
                    af[buf][i] = *reinterpret_cast<const f16x8*>(sA + lds_off(wm * WM + i * 32 + l31, ks * 2 + hi));
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                    bf[buf][j] = *reinterpret_cast<const f16x8*>(sB + lds_off(wn * WN + j * 32 + l31, ks * 2 + hi));
            } else {                                                            // rows i * 32 further on: + 4096 bytes, an immediate
                constexpr int ST = decltype(stage_c)::value;
#pragma unroll
                for (int i = 0; i < WTM; ++i)
                    af[buf][i] = *reinterpret_cast<const f16x8*>(smem + fa_off[ST / SPB][ks] + (ST % SPB) * STAGE + i * 4096);
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                    bf[buf][j] = *reinterpret_cast<const f16x8*>(smem + fb_off[ST / SPB][ks] + (ST % SPB) * STAGE + j * 4096);
            }
        };
        load_frags(0, 0);
        // SPREAD: the refill of the ring (slab it+S-1 into the stage every wave has just left) goes out a quarter behind every k-step's
        // fragment reads; its wave-uniform base is computed once, here
        const bool refill = SPREAD && !LAST && it + S - 1 < nt;
        int rst = stage + S - 1;
        if (rst >= S) rst -= S;
        ABase ab = {nullptr, 0, 0, 0};
        if (refill) ab = a_next(kt0 + it + S - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) load_frags((ks + 1) & 1, ks + 1);
            if constexpr (SPREAD) {
                if constexpr (!LAST) {
                    if (refill) {
                        a_emit(ab, rst, ks);
                        issue_w(ab.wk, rst, ks);
                    }
                } else if (ks == 1) {
                    prefetch();
                }
            } else if (LAST || !STAGGER ? ks == 1 : ks == (wave < 4 ? 1 : 3)) {
                // burst: the whole refill behind the first k-step's fragment reads (its address arithmetic overlaps matrix work);
                // STAGGER: the second wave of every SIMD two k-steps later, so that one of the two is always on the matrix pipe
                stamp(4);
                if constexpr (LAST) {
                    prefetch();
                } else if (it + S - 1 < nt) {
                    issue_aw(kt0 + it + S - 1, rst, true);
                }
                stamp(5);
            }
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)   // transposed product (weight fragment = A operand): see epi_finish
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks & 1][j], af[ks & 1][i], acc[i][j], 0, 0, 0);
            if (ks == 0) stamp(3);
        }
        stamp(6);
        if constexpr (PROF) pf_acc[0] += 1;
    };
    // Steady slabs: S - 2 younger slabs in flight behind each (S = 2: none, but the refill still follows); the last TAIL slabs find fewer
    // — counted vmcnt waits need the number at compile time, so each of them is its own copy of the slab (run-time stage).
    constexpr int TAIL = S > 2 ? S - 2 : 1, YS = S - 2;
    const int nsteady = nt > TAIL ? nt - TAIL : 0;
#ifdef SG_PIPE_RT_STAGE          // A/B build (tools/ab_lib.py): the ring stage as a run-time variable, as in rounds 1-4
    {
        int stage = 0;
        for (int it = 0; it < nsteady; ++it) {
            slab(it, std::false_type{}, std::integral_constant<int, YS>{}, stage);
            if (++stage == S) stage = 0;
        }
    }
#else
    for (int it = 0; it < nsteady; it += S) slab_seq<0, S, YS>(slab, it, nsteady);
#endif
    slab_tail<TAIL - 1, S>(slab, nt);
    if (nt <= 0) prefetch();
    if constexpr (WT == 2) epi_finish<WGM, WGN, WTM>(p, smem, acc, pre_f, pre_bias, m0, n0, z, wave, wm, wn, lane);
    else epi_finish_q<WGM, WGN>(p, smem, acc[0][0], pre_f, pre_bias, m0, n0, z, wave, lane);
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

template <int WGM, int WGN, bool CONV, int S = 3>
__global__ __launch_bounds__(64 * WGM * WGN) void mma_pipe_kernel(const MmaParams p) {
    __shared__ __attribute__((aligned(16))) char smem[pipe_smem_bytes<WGM, WGN, S>()];
    mma_pipe_body<WGM, WGN, CONV, false, S>(p, smem);
}

// The "fat wave" form (round 6): WGM x WGN waves of 128x64 on a 2-stage ring — 512x128 (4x2) and 256x256 (2x4) tiles for
// the batch-20 convolutions of the reference pass.  Per MFMA a wave issues 0.75 fragment reads and 0.25 LDS-DMA pieces (64x64 per wave:
// 1.0 and 0.375): the non-matrix instructions are what keeps the 256x128 tile at ~41 % matrix-pipe share.
template <int WGM, int WGN, bool CONV>
__global__ __launch_bounds__(64 * WGM * WGN) void mma_fat_kernel(const MmaParams p) {
    __shared__ __attribute__((aligned(16))) char smem[pipe_smem_bytes<WGM, WGN, 2, 2, 4>()];
    mma_pipe_body<WGM, WGN, CONV, false, 2, 2, 4>(p, smem);
}

// The latency form (round 6): WGM x WGN waves of 32x32 (a 64x64 tile = 4 waves), S = 4 - 8 ring stages of 16 KB.  The batch-3 main pass
// runs ~130 GEMMs per step whose FLOPs are worth 1 - 3 us and whose launches took 12 - 24 us inside the step graph (tools/bench_chain.py):
// on the 64x64-per-wave tiles a slab costs ~1 300 cycles (LDS-DMA latency / the two slabs a 3-stage ring keeps in flight; 16 MFMAs + 6 - 8
// DMA pieces per wave), and 20 - 80 of them are a dependent chain.  Here a slab is 4 MFMAs + 4 pieces per wave with up to seven slabs in
// flight, and 240 - 480 tiles fill the chip without split-K.
template <int WGM, int WGN, bool CONV, int S>
__global__ __launch_bounds__(64 * WGM * WGN) void mma_lat_kernel(const MmaParams p) {
    __shared__ __attribute__((aligned(16))) char smem[pipe_smem_bytes<WGM, WGN, S, 1>()];
    mma_pipe_body<WGM, WGN, CONV, false, S, 1>(p, smem);
}

#ifdef SG_BUILD_EXPERIMENTS
template <int WGM, int WGN, bool CONV>
__global__ __launch_bounds__(64 * WGM * WGN) void mma_pipe_prof_kernel(const MmaParams p) {
    __shared__ __attribute__((aligned(16))) char smem[pipe_smem_bytes<WGM, WGN, 3>()];
    mma_pipe_body<WGM, WGN, CONV, true>(p, smem);
}
#endif

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
        if ((int)blockIdx.x < pp.p0.tiles_m * pp.p0.tiles_n * pp.p0.splits) mma_pipe_body<WGM, WGN, false>(pp.p0, smem);
    } else {
        if ((int)blockIdx.x < pp.p1.tiles_m * pp.p1.tiles_n * pp.p1.splits) mma_pipe_body<WGM, WGN, false>(pp.p1, smem);
    }
}

template <int S>
__global__ __launch_bounds__(256) void mma_lat_pair_kernel(const MmaPair pp) {
    __shared__ __attribute__((aligned(16))) char smem[pipe_smem_bytes<2, 2, S, 1>()];
    if (blockIdx.y == 0) {
        if ((int)blockIdx.x < pp.p0.tiles_m * pp.p0.tiles_n * pp.p0.splits) mma_pipe_body<2, 2, false, false, S, 1>(pp.p0, smem);
    } else {
        if ((int)blockIdx.x < pp.p1.tiles_m * pp.p1.tiles_n * pp.p1.splits) mma_pipe_body<2, 2, false, false, S, 1>(pp.p1, smem);
    }
}

// ---- second pass of a split-K launch: partial tiles -> epilogue (8 consecutive columns per thread)
__device__ __forceinline__ void add8(float (&v)[8], const uint4& a, const uint4& b, bool f32) {
    if (f32) {
        v[0] += __uint_as_float(a.x); v[1] += __uint_as_float(a.y); v[2] += __uint_as_float(a.z); v[3] += __uint_as_float(a.w);
        v[4] += __uint_as_float(b.x); v[5] += __uint_as_float(b.y); v[6] += __uint_as_float(b.z); v[7] += __uint_as_float(b.w);
    } else {
        H8 h; h.u = a;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)h.h[e];
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
        for (int e = 0; e < 8; ++e) o.h[e] = (f16)(f32 ? __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f) : v[e]);   // see store_out4
        if (!f32) stg16(reinterpret_cast<f16*>(p.C) + (long)gm * p.ldc + gn, o.u);
        if (p.C2) stg16(p.C2 + (long)gm * p.ldc2 + gn, o.u);
    }
}

__device__ __forceinline__ void add_res8(const void* res, long ld, bool f32, int gm, int gn, float (&v)[8]) {
    if (f32) {
        const float* r = reinterpret_cast<const float*>(res) + (long)gm * ld + gn;
        add8(v, ldg16(r), ldg16(r + 4), true);
    } else {
        add8(v, ldg16(reinterpret_cast<const f16*>(res) + (long)gm * ld + gn), make_uint4(0, 0, 0, 0), false);
    }
}

__device__ __forceinline__ void epi_linear8(const MmaParams& p, int gm, int gn, float (&v)[8]) {
    if (p.bias) {
        H8 b; b.u = ldg16(p.bias + gn);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)b.h[j];
    }
    if (p.rowbias) {
        const float* rb = p.rowbias + (long)fd_div((unsigned)gm, p.fd_rpb) * p.rowbias_ld + gn;
        add8(v, ldg16(rb), ldg16(rb + 4), true);
    }
    if (p.res1) add_res8(p.res1, p.ldr1, p.flags & SG_F_RES1_F32, gm, gn, v);
    if (p.res2) add_res8(p.res2, p.ldr2, p.flags & SG_F_RES2_F32, gm, gn, v);
    store_out8(p, gm, gn, v);
    if (p.ln_out) {
        // LayerNorm fold, producer side (see epi_finish): the 8 threads of a 64-column block (consecutive lanes, same row: N % 64
        // == 0) add their sums with three exchanges, then their squared deviations from the block mean
        float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float mean = s * (1.f / 64.f);
        float m2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; m2 = fmaf(d, d, m2); }
        m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64); m2 += __shfl_xor(m2, 4, 64);
        if ((gn & 63) == 0) {
            *reinterpret_cast<float2*>(p.ln_out + ((size_t)gm * (((p.N >> 6) + 1) & ~1) + (gn >> 6)) * 2) = make_float2(s, m2);
            if (p.ln_guard && !(fabsf(mean) + sqrtf(m2) < 65504.f)) atomicOr(p.ln_guard, SG_LN_GUARD_RANGE);
        }
    }
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

// Split-K second pass that also emits the GroupNorm partial statistics of the finished rows (MmaParams::stats, same buffer
// layout as epi_finish's), so that a split-K producer does not cost its consumer a statistics pass.  One workgroup = 32 RPT rows x
// 64 columns = one partial: thread (tr = t >> 3, vc = t & 7) owns the 8-column vector vc of rows tr, tr + 32, ... (a wave reads
// 8 rows x 256 contiguous bytes per split), up to four rows x two splits in flight.  The 32 thread rows are added through LDS in a fixed
// order: deterministic, no atomics.  Host contract (check_stats / reduce_stats_rows): linear epilogue, M % (32 RPT) == 0,
// N % 64 == 0, partials never straddle two images.
template <int RPT>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const MmaParams p) {
    constexpr int RB = RPT < 4 ? RPT : 4;           // rows of a thread in flight together
    static_assert(RPT % RB == 0, "whole steps");
    __shared__ float s_acc[32][2][64];
    const int t = threadIdx.x, vc = t & 7, tr = t >> 3;
    const int nb = p.N >> 6;
    const int mt = blockIdx.x / nb, nt = blockIdx.x - mt * nb;
    const int gn = nt * 64 + vc * 8, gm0 = mt * (32 * RPT) + tr;
    const size_t MN = (size_t)p.M * p.N;
    const size_t hstep = (size_t)32 * p.N;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll 1
    for (int h0 = 0; h0 < RPT; h0 += RB) {
        float v[RB][8];
#pragma unroll
        for (int h = 0; h < RB; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[h][j] = 0.f;
        const float* s = p.ws + (size_t)(gm0 + 32 * h0) * p.N + gn;
        for (int z0 = 0; z0 < p.splits; z0 += 2) {      // partial tiles added in split order (bit-identical to the plain kernel)
            float4 a[2][RB], b[2][RB];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int h = 0; h < RB; ++h) {
                    a[u][h] = b[u][h] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (z0 + u < p.splits) {
                        const float* src = s + (z0 + u) * MN + h * hstep;
                        a[u][h] = *reinterpret_cast<const float4*>(src);
                        b[u][h] = *reinterpret_cast<const float4*>(src + 4);
                    }
                }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int h = 0; h < RB; ++h) {
                    v[h][0] += a[u][h].x; v[h][1] += a[u][h].y; v[h][2] += a[u][h].z; v[h][3] += a[u][h].w;
                    v[h][4] += b[u][h].x; v[h][5] += b[u][h].y; v[h][6] += b[u][h].z; v[h][7] += b[u][h].w;
                }
        }
#pragma unroll
        for (int h = 0; h < RB; ++h) {
            epi_linear8(p, gm0 + 32 * (h0 + h), gn, v[h]);      // bias / temb row / residuals added in place, outputs stored
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += v[h][j]; s2[j] = fmaf(v[h][j], v[h][j], s2[j]); }
        }
    }
    *reinterpret_cast<float4*>(&s_acc[tr][0][vc * 8]) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    *reinterpret_cast<float4*>(&s_acc[tr][0][vc * 8 + 4]) = make_float4(s1[4], s1[5], s1[6], s1[7]);
    *reinterpret_cast<float4*>(&s_acc[tr][1][vc * 8]) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    *reinterpret_cast<float4*>(&s_acc[tr][1][vc * 8 + 4]) = make_float4(s2[4], s2[5], s2[6], s2[7]);
    __syncthreads();
    if (t < 128) {
        const int plane = t >> 6, ch = t & 63;
        float acc = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) acc += s_acc[r][plane][ch];
        p.stats[((size_t)mt * 2 + plane) * p.N + nt * 64 + ch] = acc;
    }
}

// Rows per statistics partial of a split-K launch (0 = it cannot emit them): the largest of 256 / 128 / 64 that divides an image
// and still leaves the second pass a workgroup for most CUs (fewer, larger partials make the consumer's merge shorter, but a
// 120-workgroup second pass measured 14 us against 7 us for the plain one — profiles/r03e_kernel_stats.csv).
int reduce_stats_rows(const MmaParams& p) {
    if (p.mode != SG_EPI_LINEAR || p.N % 64 != 0 || p.stats_batch_rows <= 0 || p.M % p.stats_batch_rows != 0) return 0;
    int best = 0;
    for (int rows = 64; rows <= 256; rows *= 2)
        if (p.stats_batch_rows % rows == 0 && (best == 0 || (long)(p.M / rows) * (p.N / 64) >= 192)) best = rows;
    return best;
}

// ------------------------------------------------------------------------------------------------ host side
struct Plan { int bm, bn, splits; int lat; int fat = 0; };      // lat: 0 = 64x64 per wave; else the 32x32-per-wave kernel (mma_lat_kernel) and its ring depth

// Development options: storygen_amd/csrc/common.h SgOptions (set through sg_debug_set_option; never from the environment).
struct TuneView {
    SgOptions& o = sg_options();
    int& bm = o.tile_m; int& bn = o.tile_n; int& no_pipe = o.no_pipe; int& no_split = o.no_split; int& no_nmajor = o.no_nmajor;
};
static const TuneView g_tune;

// Cost model (cycles at ~2.4 GHz).  Measured on MI355X (tools/bench_gemm.py): a CU pulls operand slabs from L2 into
// LDS at ~18.5 B/cycle however many waves ask (L1 miss-level parallelism x L2 latency), so a launch is bound by
//   t_bw   = (blocks a CU must run) x (bytes one block streams) / 18.5      — total bytes over the ACTIVE CUs, or by
//   t_mfma = (waves per SIMD) x slabs x 512                                  — 16 MFMAs of 32 cycles per 64-deep slab,
// plus a fixed prologue/epilogue.  Splitting K does not add operand bytes but multiplies the CUs that share them,
// which is what small-M layers need; it costs a second launch that re-reads the fp32 partial tiles.
Plan choose_plan(int M, int N, int KT, int force_split, int max_ws_split, bool pipe, int hint_bm, int hint_bn) {
    static const int cand_pipe[6][2] = {{256, 128}, {128, 128}, {256, 64}, {128, 64}, {64, 128}, {64, 64}};
    static const int cand_gen[1][2] = {{128, 128}};      // the register-staged kernel is a cold path: one tile shape
    static const int split_opts[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
    const int ncand = pipe ? 6 : 1;
    const double CUS = 256.0, BW = pipe ? 18.5 : 12.0;
    Plan best{64, 64, 1, 0};
    double best_cost = 1e300;
    for (int ci = 0; ci < ncand; ++ci) {
        const int bm = pipe ? cand_pipe[ci][0] : cand_gen[ci][0], bn = pipe ? cand_pipe[ci][1] : cand_gen[ci][1];
        if (pipe && g_tune.bm && (bm != g_tune.bm || bn != g_tune.bn)) continue;
        if (pipe && !g_tune.bm && hint_bm && (bm != hint_bm || bn != hint_bn)) continue;
        const long tiles = (long)sg_cdiv(M, bm) * sg_cdiv(N, bn);
        const double waves_per_block = pipe ? (bm / 64) * (bn / 64) : 4.0;
        const double mfma_per_slab = pipe ? 512.0 : 512.0 * (bm / 64.0) * (bn / 64.0) / 4.0;
        for (int s : split_opts) {
            if (force_split > 0 && s != force_split) continue;
            if (force_split <= 0 && s > 1 && (s > max_ws_split || KT / s < 2 || g_tune.no_split)) continue;
            if (s > KT) continue;
            const double blocks = (double)tiles * s;
            const double slabs = sg_cdiv(KT, s);
            const double blocks_per_cu = sg_cdiv((long)blocks, (long)CUS);
            const double t_bw = blocks_per_cu * slabs * (bm + bn) * 128.0 / BW;
            const double waves_per_simd = sg_cdiv((long)(blocks * waves_per_block), (long)(CUS * 4));
            const double t_mfma = waves_per_simd * slabs * mfma_per_slab;
            double cost = (t_bw > t_mfma ? t_bw : t_mfma) + 2500.0 + blocks_per_cu * (bm * bn / 16.0);
            if (s > 1) cost += 5000.0 + (double)M * N * 4.0 * (s + 1) / 1500.0;   // second launch + partial tiles
            if (cost < best_cost) { best_cost = cost; best = Plan{bm, bn, s, 0}; }
        }
    }
    if (best_cost == 1e300) best = Plan{64, 64, force_split > KT ? KT : (force_split > 0 ? force_split : 1), 0};
    return best;
}

template <int WGM, int WGN, bool CONV>
void launch_pipe(const MmaParams& p, dim3 grid, hipStream_t st) {
    const dim3 block(64 * WGM * WGN);
#ifdef SG_BUILD_EXPERIMENTS
    if (p.prof) {
        hipLaunchKernelGGL((mma_pipe_prof_kernel<WGM, WGN, CONV>), grid, block, 0, st, p);
        return;
    }
#endif
    // development option pipe_stages = 2: the tiles whose 3-stage ring keeps a second workgroup off the CU run a 2-stage ring
    if constexpr ((WGM == 2 && WGN == 2) || (WGM == 4 && WGN == 1)) {
        if (sg_options().pipe_stages == 2) {
            hipLaunchKernelGGL((mma_pipe_kernel<WGM, WGN, CONV, 2>), grid, block, 0, st, p);
            return;
        }
    }
    hipLaunchKernelGGL((mma_pipe_kernel<WGM, WGN, CONV>), grid, block, 0, st, p);
}

thread_local unsigned long long* g_prof = nullptr;     // set by sg_debug_*_anatomy around one launch
thread_local int g_query_rows = 0;                     // result of a stats query (rows per partial = the tile height), 0 = none
thread_local bool g_stats_query = false;               // sg_*_stats_tile_rows: plan only, report eligibility instead of failing
thread_local bool g_plan_query = false;                // sg_conv3x3_planned_splits: plan only, report the number of K slices
thread_local int32_t* g_plan_out = nullptr;            // sg_*_launch_plan: also {tile rows, tile columns, K slices, workgroups, threads per workgroup, pipelined}

// GroupNorm statistics from the epilogue (MmaParams::stats) need whole tiles inside one image and the linear epilogue; a split-K
// launch emits them from its second pass.  A launch that was asked for them but cannot deliver fails (the caller asks
// sg_*_stats_tile_rows first).
int check_stats(MmaParams& p, int bm, const char* name) {
    if (!p.stats) return SG_OK;
    // fused epilogue: one partial per row tile of the launch; split-K: from the second pass (reduce_stats_rows)
    const int rows = p.splits == 1 ? bm : reduce_stats_rows(p);
    const bool ok = p.mode == SG_EPI_LINEAR && rows > 0 && p.stats_batch_rows > 0 && p.stats_batch_rows % rows == 0 &&
                    p.M % p.stats_batch_rows == 0;
    if (ok) return SG_OK;
    if (g_stats_query) { p.stats = nullptr; return SG_OK; }
    return sg_set_error(SG_EINVAL, "%s: epilogue statistics need the linear epilogue, a %d-row tile that divides the %d rows of an image "
                        "and, under split-K (here %d), N %% 64 == 0: query sg_*_stats_tile_rows first", name, rows, p.stats_batch_rows,
                        p.splits);
}

// Decomposition of one problem: tile shape, K split, tile order; fills the corresponding fields of p.  `pipe` = the LDS-DMA
// kernel applies (no load needs a predicate: K % 64 == 0; conv input zero-bordered), else the register-staged kernel.
// The 32x32-per-wave kernels (mma_lat_kernel): for launches that are a short dependent chain rather than a volume of FLOPs.
//   64x64 tile, four waves, 4-stage ring: GEMMs of <= lat_tiles tiles and lat_min_kt .. lat_max_kt K slabs (development options);
//   64x128 tile, eight waves, 6-stage ring (80 KB of WEIGHTS in flight per workgroup): M <= 256 — the 8x8 level, where a launch is its
//   weight stream (29.5 MB for a 1280 -> 1280 convolution against 0.5 MB of activations) and what bounds it is the bytes a CU keeps in
//   flight (the 64x64-per-wave tiles: 16 - 32 KB).
// hint_waves = 4 with a 64x64 hint / 8 with 64x128: the caller asks for one; -1: never; 0: by size.  Fills pl (tile, K slices, ring
// depth) and returns true, or returns false when the launch stays on the 64x64-per-wave kernels.
thread_local bool g_in_pair = false;        // sg_gemm_pair_f16 is planning (development option lat_mask)

template <bool CONV>
bool lat_plan(const MmaParams& p, bool pipe, int force_split, int max_ws_split, int hint_bm, int hint_bn, int hint_waves, Plan& pl) {
    if (!pipe || p.mode != SG_EPI_LINEAR || p.prof || hint_waves < 0 || g_tune.bm) return false;
    const SgOptions& o = sg_options();
    // development option lat_mask (bisecting): which launch kinds may take the kernel by size — 1 paired launches, 2 LayerNorm-folded
    // consumers, 4 GroupNorm partials, 8 K slices, 16 LayerNorm-partial producers, 32 everything else
    if (hint_waves == 0) {
        // (the columns-are-tokens fold, ln_mode 2, only ever occurs as the second problem of a pair: it shares the pairs' bit)
        const int kind = (g_in_pair || p.ln_mode == 2) ? 1 : p.ln_mode ? 2 : p.stats ? 4 : p.ln_out ? 16 : 32;
        if (!(o.lat_mask & kind)) return false;
    }
    const bool hinted = hint_bm != 0 || hint_bn != 0 || hint_waves != 0;
    const bool ask_sq = hint_bm == 64 && hint_bn == 64 && hint_waves == 4, ask_wide = hint_bm == 64 && hint_bn == 128 && hint_waves == 8;
    if (hinted && !ask_sq && !ask_wide) return false;
    const long tiles_sq = (long)sg_cdiv(p.M, 64) * sg_cdiv(p.N, 64), tiles_wide = (long)sg_cdiv(p.M, 64) * sg_cdiv(p.N, 128);
    const bool wide = ask_wide || (!hinted && o.lat_wide && p.M <= o.lat_wide_m && p.N >= 256 && p.KT >= 40);
    const bool sq = !wide && (ask_sq || (!hinted && !CONV && tiles_sq <= o.lat_tiles && p.KT >= o.lat_min_kt &&
                                         (p.KT <= o.lat_max_kt || tiles_sq <= 128)));
    if (!wide && !sq) return false;
    const long tiles = wide ? tiles_wide : tiles_sq;
    // K slices only where the tiles alone leave most CUs idle (M <= 256 at N = 1280): enough for ~256 workgroups, >= min_slabs each
    const int min_slabs = wide ? 8 : 16;
    int sp = 1;
    if (force_split > 0) sp = force_split > p.KT ? p.KT : force_split;
    else if (tiles <= 128 && !g_tune.no_split && (o.lat_mask & 8)) {
        sp = (int)(256 / tiles);
        if (sp > p.KT / min_slabs) sp = p.KT / min_slabs;
        if (sp > max_ws_split) sp = max_ws_split;
        if (sp > MAX_AUTO_SPLIT) sp = MAX_AUTO_SPLIT;
        if (sp < 1) sp = 1;
    }
    // 64x64: 4 stages (three slabs = 48 KB in flight per workgroup, two workgroups per CU) measured equal or better than 8 on every
    // main-pass shape, incl. those of <= 256 workgroups (M768 N1280 K1280: 9.9 vs 10.7 us per graph node — the 8-stage prologue issues
    // 28 DMA pieces per wave before the first slab can land; profiles/r06b_*); 8 stays behind the development option
    pl = Plan{64, wide ? 128 : 64, sp, wide ? 6 : (o.lat_stages == 8 ? 8 : 4)};
    return true;
}

template <bool CONV>
int plan_mma(MmaParams& p, int force_split, int hint_bm, int hint_bn, int hint_waves, void* ws, size_t ws_bytes, const char* name,
             Plan& pl, bool& pipe) {
    p.KT = sg_cdiv(p.K, BK);
    p.prof = g_prof;
    pipe = (p.K % BK == 0) && (!CONV || p.padded) && !g_tune.no_pipe;
    const size_t per_split = (size_t)p.M * p.N * 4;
    const int max_ws_split = ws ? (int)(ws_bytes / per_split > 64 ? 64 : ws_bytes / per_split) : 1;
    if (sg_options().big_m > 0 && p.M >= sg_options().big_m && hint_bm == 0 && hint_bn == 0 && hint_waves == 0 && pipe) {
        hint_bm = sg_options().big_bm; hint_bn = sg_options().big_bn;       // development option: see common.h
    }
    const bool ask_fat = hint_waves == 8 && ((hint_bm == 512 && hint_bn == 128) || (hint_bm == 256 && hint_bn == 256));
    const bool fat_ok = pipe && !p.prof && !g_tune.bm && force_split <= 1 && (p.mode == SG_EPI_LINEAR || p.mode == SG_EPI_GEGLU);
    // development option fat_m: large convolutions without a hint take 512x128 where an image's rows divide by 512 (GroupNorm partials are
    // per row tile)
    int auto_bm = 0, auto_bn = 0;
    if (CONV && fat_ok && sg_options().fat_m > 0 && p.M >= sg_options().fat_m && hint_bm == 0 && hint_bn == 0 && hint_waves == 0) {
        // (256x256 lost to 512x128 on every shape measured, profiles/r06bi_*: by hint only)
        const int rows = p.stats ? p.stats_batch_rows : 512;
        if (rows % 512 == 0) { auto_bm = 512; auto_bn = 128; }
    }
    if (ask_fat && fat_ok) {
        pl = Plan{hint_bm, hint_bn, 1, 0, 1};
    } else if (auto_bm) {
        pl = Plan{auto_bm, auto_bn, 1, 0, 1};
    } else if (!lat_plan<CONV>(p, pipe, force_split, max_ws_split, ask_fat ? 0 : hint_bm, ask_fat ? 0 : hint_bn, ask_fat ? 0 : hint_waves, pl)) {
        if (ask_fat) hint_bm = hint_bn = 0;                                                 // asked for, not applicable: the cost model decides
        if ((hint_waves == 4 && hint_bm == 64 && hint_bn == 64) || (hint_waves == 8 && hint_bm == 64 && hint_bn == 128))
            hint_bm = hint_bn = 0;                                                          // asked for, not applicable: the cost model decides
        pl = choose_plan(p.M, p.N, p.KT, force_split, max_ws_split, pipe, hint_bm, hint_bn);
    }
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
    // group_m: the number of row tiles a wave of ~32 / splits consecutive tiles spans (tile_of_id).  Bytes that wave asks its L2 for =
    // gm activation panels + (wave / gm) weight panels; pick the gm in {1, 2, 4, 8, 16, all} with the fewest (ties: the larger gm, i.e.
    // the former M-first walk).  Development option no_nmajor = 1: N first everywhere (group_m = 1).
    {
        const double a_panel = a_bytes / p.tiles_m, w_panel = w_bytes / p.tiles_n;
        const int wave_tiles = pl.splits >= 32 ? 1 : 32 / pl.splits;
        int best = 1;
        double best_bytes = 1e300;
        const int cands[6] = {1, 2, 4, 8, 16, p.tiles_m};
        for (int c : cands) {
            const int gm = c < p.tiles_m ? c : p.tiles_m;
            const int width = gm * p.tiles_n;
            int rows, cols;
            if (width >= wave_tiles) {                     // the wave lies inside one group: gm rows x wave / gm columns
                rows = gm < wave_tiles ? gm : wave_tiles;
                cols = (wave_tiles + rows - 1) / rows;
            } else {                                       // it spans several whole groups: all columns, gm rows per group
                rows = gm * ((wave_tiles + width - 1) / width);
                if (rows > p.tiles_m) rows = p.tiles_m;
                cols = p.tiles_n;
            }
            const double bytes = rows * a_panel + cols * w_panel;
            if (bytes <= best_bytes) { best_bytes = bytes; best = gm; }
        }
        p.group_m = g_tune.no_nmajor ? 1 : best;
    }
    p.fd_splits = make_fastdiv((unsigned)p.splits);
    p.fd_group_w = make_fastdiv((unsigned)(p.group_m * p.tiles_n));
    p.fd_group_m = make_fastdiv((unsigned)p.group_m);
    p.fd_rpb = make_fastdiv((unsigned)(p.rows_per_batch > 0 ? p.rows_per_batch : 1));
    if (CONV) {
        p.fd_hw = make_fastdiv((unsigned)(p.Ho * p.Wo));
        p.fd_wo = make_fastdiv((unsigned)p.Wo);
        p.fd_cpt = make_fastdiv((unsigned)p.cpt);
    }
    return check_stats(p, pl.bm, name);
}

int launch_reduce(const MmaParams& p, hipStream_t st) {
    if (p.splits <= 1) return SG_OK;
    if (p.stats) {      // check_stats made sure of the contract
        const int rows = reduce_stats_rows(p);
        const dim3 grid((p.M / rows) * (p.N / 64));
        if (rows == 256) hipLaunchKernelGGL(splitk_reduce_stats_kernel<8>, grid, dim3(256), 0, st, p);
        else if (rows == 128) hipLaunchKernelGGL(splitk_reduce_stats_kernel<4>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(splitk_reduce_stats_kernel<2>, grid, dim3(256), 0, st, p);
        SG_CHECK_LAUNCH("splitk_reduce_stats");
        return SG_OK;
    }
    const long items = (long)p.M * (p.N / (p.mode == SG_EPI_GEGLU ? 16 : 8));
    const int blocks = (int)min((long)4096, (items + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    SG_CHECK_LAUNCH("splitk_reduce");
    return SG_OK;
}

template <bool CONV>
int launch_mma(MmaParams& p, int force_split, int hint_bm, int hint_bn, int hint_waves, void* ws, size_t ws_bytes, hipStream_t st, const char* name) {
    Plan pl;
    bool pipe;
    if (int rc = plan_mma<CONV>(p, force_split, hint_bm, hint_bn, hint_waves, ws, ws_bytes, name, pl, pipe)) return rc;
    if (g_stats_query) {
        g_query_rows = p.stats ? (pl.splits > 1 ? reduce_stats_rows(p) : pl.bm) : 0;
        return SG_OK;
    }
    if (g_plan_query) {
        g_query_rows = pl.splits;
        if (g_plan_out) {
            g_plan_out[0] = pl.bm; g_plan_out[1] = pl.bn; g_plan_out[2] = pl.splits; g_plan_out[3] = p.tiles_m * p.tiles_n * pl.splits;
            g_plan_out[4] = pl.fat ? 512 : pl.lat ? 64 * (pl.bm / 32) * (pl.bn / 32) : pipe ? 64 * (pl.bm / 64) * (pl.bn / 64) : 256;
            g_plan_out[5] = pl.fat ? 2 : pl.lat ? 16 + pl.lat : pipe ? 1 : 0;        // 2: 128x64 per wave; 16 + ring depth: the 32x32-per-wave kernel
        }
        return SG_OK;
    }
    if (p.defer && pl.splits <= 1)
        return sg_set_error(SG_EINVAL, "%s: defer_reduce needs a split-K launch (query sg_conv3x3_planned_splits first)", name);
    dim3 grid(p.tiles_m * p.tiles_n * pl.splits);
    if (pl.fat) {
        if (pl.bm == 512) hipLaunchKernelGGL((mma_fat_kernel<4, 2, CONV>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((mma_fat_kernel<2, 4, CONV>), grid, dim3(512), 0, st, p);
    } else if (pl.lat) {
        if (pl.bn == 128) hipLaunchKernelGGL((mma_lat_kernel<2, 4, CONV, 6>), grid, dim3(512), 0, st, p);
        else if (pl.lat == 8) hipLaunchKernelGGL((mma_lat_kernel<2, 2, CONV, 8>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((mma_lat_kernel<2, 2, CONV, 4>), grid, dim3(256), 0, st, p);
    } else if (pipe) {
        if (pl.bm == 256 && pl.bn == 128) launch_pipe<4, 2, CONV>(p, grid, st);
        else if (pl.bm == 128 && pl.bn == 128) launch_pipe<2, 2, CONV>(p, grid, st);
        else if (pl.bm == 256 && pl.bn == 64) launch_pipe<4, 1, CONV>(p, grid, st);
        else if (pl.bm == 128 && pl.bn == 64) launch_pipe<2, 1, CONV>(p, grid, st);
        else if (pl.bm == 64 && pl.bn == 128) launch_pipe<1, 2, CONV>(p, grid, st);
        else launch_pipe<1, 1, CONV>(p, grid, st);
    } else {
        dim3 block(256);
        if (p.ln_mode) return sg_set_error(SG_EINVAL, "%s: a folded LayerNorm needs the LDS-DMA kernel (K %% 64 == 0)", name);
        if (pl.bm == 128 && pl.bn == 128) hipLaunchKernelGGL((mma_kernel<128, 128, CONV>), grid, block, 0, st, p);
        else return sg_set_error(SG_EINVAL, "%s: internal: the register-staged kernel has only the 128x128 tile", name);
    }
    SG_CHECK_LAUNCH(name);
    if (p.defer) return SG_OK;          // the consumer sums the slices (sg_groupnorm_desc.split_ws)
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
    if ((bm == 64 && bn == 64 && waves == 4) || (bm == 64 && bn == 128 && waves == 8)) return SG_OK;       // the 32x32-per-wave kernels (mma_lat_kernel)
    if (bm == 0 && bn == 0 && waves == -1) return SG_OK;                                                   // heuristic tile, never the latency kernel
    if (waves == 8 && ((bm == 512 && bn == 128) || (bm == 256 && bn == 256))) return SG_OK;                // eight waves of 128x64 (mma_fat_kernel)
    if (waves != 0 && waves != (bm / 64) * (bn / 64))
        return sg_set_error(SG_EINVAL, "%s: tile_waves=%d is not available for tile %dx%d (64x64 per wave; 64x64 with 4 waves of 32x32)", who, waves, bm, bn);
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
    SG_REQUIRE(d->ln_mode >= 0 && d->ln_mode <= 2, "%s: ln_mode must be 0, 1 or 2", who);
    if (d->ln_mode) {
        SG_REQUIRE(d->ln_stats && d->ln_c && d->ln_d, "%s: ln_mode needs ln_stats, ln_c and ln_d", who);
        SG_REQUIRE(d->ln_parts >= 1 && d->ln_parts <= LN_MAX_PARTS && d->K == 64 * d->ln_parts,
                   "%s: ln_parts (%d) must be K / 64 (K = %d) and at most %d", who, d->ln_parts, d->K, LN_MAX_PARTS);
        SG_REQUIRE(sg_aligned16(d->ln_c) && sg_aligned16(d->ln_d) && sg_aligned16(d->ln_stats), "%s: ln_c / ln_d / ln_stats alignment", who);
        SG_REQUIRE(d->split_k <= 1, "%s: a GEMM with a folded LayerNorm does not split K (its epilogue is not linear in the partials)", who);
        SG_REQUIRE(!d->stats && !d->rowbias && !d->res1 && !d->res2, "%s: ln_mode excludes stats, rowbias and residuals", who);
        SG_REQUIRE(d->ln_mode == 1 || d->epilogue == SG_EPI_LINEAR, "%s: ln_mode 2 needs the linear epilogue", who);
        SG_REQUIRE(d->ln_eps > 0.f, "%s: ln_eps must be positive", who);
    }
    if (d->ln_stats_out) {
        SG_REQUIRE(d->epilogue == SG_EPI_LINEAR && d->N % 64 == 0 && sg_aligned16(d->ln_stats_out),
                   "%s: ln_stats_out needs the linear epilogue, N %% 64 == 0 and a 16-byte aligned buffer", who);
    }
    p.ln_stats = d->ln_stats; p.ln_parts = d->ln_parts; p.ln_c = d->ln_c; p.ln_d = d->ln_d; p.ln_mode = d->ln_mode; p.ln_eps = d->ln_eps;
    p.ln_out = d->ln_stats_out;
    SG_REQUIRE(!d->ln_guard || (reinterpret_cast<uintptr_t>(d->ln_guard) & 3u) == 0, "%s: ln_guard must be 4-byte aligned", who);
    p.ln_guard = (d->ln_mode || d->ln_stats_out) ? reinterpret_cast<unsigned*>(d->ln_guard) : nullptr;
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
    return launch_mma<false>(p, d->ln_mode ? 1 : d->split_k, d->tile_m, d->tile_n, d->tile_waves, d->workspace, d->workspace_bytes, (hipStream_t)stream, "sg_gemm_f16");
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
    const int sk0 = d0->ln_mode ? 1 : d0->split_k, sk1 = d1->ln_mode ? 1 : d1->split_k;
    struct PairScope { PairScope() { g_in_pair = true; } ~PairScope() { g_in_pair = false; } } pair_scope;
    if (int rc = plan_mma<false>(pp.p0, sk0, d0->tile_m, d0->tile_n, d0->tile_waves, d0->workspace, d0->workspace_bytes, "sg_gemm_pair_f16[0]", pl0, pipe0)) return rc;
    // one kernel instantiation serves both problems: the second one is planned on the first one's tile shape (and kernel family)
    if (int rc = plan_mma<false>(pp.p1, sk1, pl0.bm, pl0.bn, pl0.lat ? (pl0.bn == 128 ? 8 : 4) : -1, d1->workspace, d1->workspace_bytes, "sg_gemm_pair_f16[1]", pl1, pipe1)) return rc;
    if (!pipe0 || !pipe1 || pl1.bm != pl0.bm || pl1.bn != pl0.bn || (pl0.lat != 0) != (pl1.lat != 0) || (pl0.lat && pl0.bn == 128)) {        // not pairable (K % 64, forced tile): two launches
        if (int rc = launch_mma<false>(pp.p0, sk0, d0->tile_m, d0->tile_n, d0->tile_waves, d0->workspace, d0->workspace_bytes, st, "sg_gemm_pair_f16[0]")) return rc;
        return launch_mma<false>(pp.p1, sk1, d1->tile_m, d1->tile_n, d1->tile_waves, d1->workspace, d1->workspace_bytes, st, "sg_gemm_pair_f16[1]");
    }
    const int g0 = pp.p0.tiles_m * pp.p0.tiles_n * pp.p0.splits, g1 = pp.p1.tiles_m * pp.p1.tiles_n * pp.p1.splits;
    // grid.x is a multiple of 8 so that block (x, 1) sits on XCD x % 8 like block (x, 0): xcd_remap keeps its meaning
    dim3 grid(((g0 > g1 ? g0 : g1) + 7) & ~7, 2);
    if (pl0.lat) {
        if (sg_options().lat_stages == 8) hipLaunchKernelGGL(mma_lat_pair_kernel<8>, grid, dim3(256), 0, st, pp);
        else hipLaunchKernelGGL(mma_lat_pair_kernel<4>, grid, dim3(256), 0, st, pp);
    }
    else if (pl0.bm == 256 && pl0.bn == 128) launch_pair<4, 2>(pp, grid, st);
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
    SG_REQUIRE(9 * (d->Cin / 64) < 3000, "sg_conv3x3: Cin (%d) too large for the slab decode (kt / 9 by multiply-shift, kt < 3000)", d->Cin);
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
    SG_REQUIRE(d->defer_reduce == 0 || (d->defer_reduce == 1 && !d->stats), "sg_conv3x3: defer_reduce is 0 or 1 and excludes stats");
    p.defer = d->defer_reduce;
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

extern "C" int sg_conv3x3_planned_splits(const sg_conv3x3_desc* d) {
    SG_REQUIRE(d != nullptr, "sg_conv3x3_planned_splits: null descriptor");
    sg_conv3x3_desc q = *d;
    q.defer_reduce = 0;
    g_plan_query = true; g_query_rows = 0;
    const int rc = sg_conv3x3_nhwc_f16(&q, nullptr);
    g_plan_query = false;
    return rc ? rc : g_query_rows;
}

// The decomposition a launch with this descriptor will use (host-only, no launch): out[6] = {tile rows, tile columns, K slices,
// workgroups, threads per workgroup, 1 if the LDS-DMA kernel applies}.  Measurement tooling joins it with a profiler's (kernel,
// grid) classes (tools/traffic_from_pmc.py: algorithmic bytes per class).
extern "C" int sg_gemm_launch_plan(const sg_gemm_desc* d, int32_t* out) {
    SG_REQUIRE(d && out, "sg_gemm_launch_plan: null argument");
    g_plan_query = true; g_query_rows = 0; g_plan_out = out;
    const int rc = sg_gemm_f16(d, nullptr);
    g_plan_query = false; g_plan_out = nullptr;
    return rc;
}

extern "C" int sg_conv3x3_launch_plan(const sg_conv3x3_desc* d, int32_t* out) {
    SG_REQUIRE(d && out, "sg_conv3x3_launch_plan: null argument");
    sg_conv3x3_desc q = *d;
    q.defer_reduce = 0;
    g_plan_query = true; g_query_rows = 0; g_plan_out = out;
    const int rc = sg_conv3x3_nhwc_f16(&q, nullptr);
    g_plan_query = false; g_plan_out = nullptr;
    return rc;
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
// kernel; prof receives 10 uint64 per wave ([block][wave][10], see mma_pipe_body).  Only in a library built with
// SG_BUILD_EXPERIMENTS=1 (python -m storygen_amd.build --experiments); the product library answers SG_EINVAL.
extern "C" int sg_debug_gemm_anatomy(const sg_gemm_desc* d, void* prof, size_t prof_bytes, sg_stream_t stream) {
#ifdef SG_BUILD_EXPERIMENTS
    SG_REQUIRE(d && prof && prof_bytes >= (size_t)8 * 10 * 8 * 65536 / 64, "sg_debug_gemm_anatomy: need a profile buffer (>= 80 B per wave)");
    g_prof = reinterpret_cast<unsigned long long*>(prof);
    const int rc = sg_gemm_f16(d, stream);
    g_prof = nullptr;
    return rc;
#else
    (void)d; (void)prof; (void)prof_bytes; (void)stream;
    return sg_set_error(SG_EINVAL, "sg_debug_gemm_anatomy: this library was built without SG_BUILD_EXPERIMENTS");
#endif
}

extern "C" int sg_debug_conv_anatomy(const sg_conv3x3_desc* d, void* prof, size_t prof_bytes, sg_stream_t stream) {
#ifdef SG_BUILD_EXPERIMENTS
    SG_REQUIRE(d && prof && prof_bytes >= (size_t)8 * 10 * 8 * 65536 / 64, "sg_debug_conv_anatomy: need a profile buffer (>= 80 B per wave)");
    g_prof = reinterpret_cast<unsigned long long*>(prof);
    const int rc = sg_conv3x3_nhwc_f16(d, stream);
    g_prof = nullptr;
    return rc;
#else
    (void)d; (void)prof; (void)prof_bytes; (void)stream;
    return sg_set_error(SG_EINVAL, "sg_debug_conv_anatomy: this library was built without SG_BUILD_EXPERIMENTS");
#endif
}

extern "C" int sg_debug_set_tile(int32_t bm, int32_t bn, int32_t no_pipe) {
    g_tune.bm = bm; g_tune.bn = bn; g_tune.no_pipe = no_pipe;
    return SG_OK;
}

// host-side check of the multiply-shift divisors (tests/test_abi.py): returns the number of mismatches over a sweep
extern "C" int sg_debug_fastdiv_selftest(void) {
    int bad = 0;
    const unsigned ds[] = {1, 2, 3, 5, 7, 9, 10, 20, 40, 45, 64, 96, 144, 1024, 2304, 4096, 9216, 36864, 65535, 65536, 1000003};
    for (unsigned d : ds) {
        const FastDiv f = make_fastdiv(d);
        for (unsigned long long n = 0; n < (1ull << 32); n += 65521ull) bad += fd_div((unsigned)n, f) != (unsigned)n / d;
        for (unsigned n = 0; n < 200000; ++n) bad += fd_div(n, f) != n / d;
        bad += fd_div(0xFFFFFFFFu, f) != 0xFFFFFFFFu / d;
    }
    // the grouped tile order (tile_of_id) is a bijection onto the tile grid for every group size, incl. a smaller last group
    for (int tiles_m : {1, 2, 3, 5, 8, 20, 33})
        for (int tiles_n : {1, 3, 10, 80})
            for (int gm : {1, 2, 4, 7, 8, 16, 64}) {
                if (gm > tiles_m) gm = tiles_m;
                const FastDiv fw = make_fastdiv((unsigned)(gm * tiles_n)), fg = make_fastdiv((unsigned)gm);
                std::vector<char> seen((size_t)tiles_m * tiles_n, 0);
                for (unsigned lid = 0; lid < (unsigned)(tiles_m * tiles_n); ++lid) {
                    unsigned tm, tn;
                    tile_of_id(lid, tiles_m, tiles_n, gm, fw, fg, tm, tn);
                    if (tm >= (unsigned)tiles_m || tn >= (unsigned)tiles_n || seen[(size_t)tm * tiles_n + tn]) ++bad;
                    else seen[(size_t)tm * tiles_n + tn] = 1;
                }
            }
    // the convolution's slab decode: kt / 9 by multiply-shift
    for (int kt = 0; kt < 3000; ++kt) bad += ((kt * 7282) >> 16) != kt / 9;
    return bad;
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
