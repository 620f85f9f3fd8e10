// Shared device/host helpers for the gfx950 kernels of libstorygen_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/storygen_hip.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------- host side
int sg_set_error(int code, const char* fmt, ...);

#define SG_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return sg_set_error(SG_EINVAL, __VA_ARGS__); \
    } while (0)

#define SG_CHECK_LAUNCH(name)                                                                  \
    do {                                                                                       \
        hipError_t e__ = hipGetLastError();                                                    \
        if (e__ != hipSuccess) return sg_set_error(SG_ELAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline bool sg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int sg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Development options (kernel-variant selectors used by the tuning / anatomy tools and the parity tests).  ONE process-global
// table, defined in misc.hip, written only through sg_debug_set_option(): the library never reads the environment, so a
// product run cannot be altered by a stray variable (tools map SG_* environment variables onto it in Python, storygen_amd/ops.py).
struct SgOptions {
    int tile_m = 0, tile_n = 0;        // force a GEMM / conv tile (0, 0 = heuristic / caller's hint)
    int no_pipe = 0, no_split = 0;     // register-staged kernel only / no automatic split-K
    int no_nmajor = 0;                 // M-major tile order everywhere
    int pipe_stages = 3;               // 2: 128x128 and 256x64 GEMM / conv tiles on a 2-stage LDS ring (two workgroups per CU)
    int lat_tiles = 640;               // GEMMs of at most this many 64x64 tiles run the 32x32-per-wave deep-ring kernel (mma_lat_kernel); 0 = only on a caller's hint
    int lat_min_kt = 8, lat_max_kt = 64;   // ... with this many 64-deep K slabs
    int lat_stages = 4;                // ring depth of that kernel: 4 (default) or 8
    int lat_mask = 62;                 // launch kinds that may take it by size: 1 paired launches (OFF: with pairs on it the one-graph loop differed run to run
                                       // in 6 - 12 of 30 repeats, profiles/r06h_*; every other kind 30 / 30 bit-identical), 2 LN-folded consumers, 4 GroupNorm
                                       // partials, 8 K slices, 16 LN-partial producers, 32 others
    int big_m = 0, big_bm = 0, big_bn = 0; // big_m > 0: launches of M >= big_m rows without a tile hint take the (big_bm, big_bn) tile — the batched reference pass on smaller workgroups (A/B: how long a CU is held matters to the co-running main pass)
    int fat_m = 0;                         // > 0: convolutions of M >= fat_m rows take the 128x64-per-wave tiles (mma_fat_kernel: 512x128 / 256x256) when no tile is hinted
    int lat_wide = 0, lat_wide_m = 256;   // 1: M <= lat_wide_m (the 8x8 level) takes the 64x128-tile / 6-stage weight-streaming form (measured neutral: default off; tile hint (64, 128, 8) selects it per launch)
    int attn_sub2 = 0, attn_prio = 0, attn_d80 = 1, attn_d160 = 4 /* 4: key-split workgroups at Nq <= 256 */, attn_lean = 0;
    int attn_d40_general = 0;          // 1 = the D = 40 launches use the general softmax path (A/B against the padded-dimension fast path)
    int gn_no_fused = 0, gn_wide = 1;
    int gn_fused_nt = 1024;            // threads of the one-launch GroupNorm for slabs <= 16 384 values (256: the round-1 geometry; A/B)
    int gn_chunks = 0;                 // wide GroupNorm: cap on the row chunks per sample (0 = 256 / B, one round of workgroups; 64 = rounds 1-3)
    int ff_variant = 3;                // fused feed-forward: bit 0 = refill spread over the k-steps, bit 1 = fragments two k-steps ahead
    long gn_fused_max = -1;            // -1 = the kernel's default threshold
};
SgOptions& sg_options();

// ---------------------------------------------------------------------------------------------- device side
__device__ __forceinline__ uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void stg16(void* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

union H8 {
    uint4 u;
    f16x8 v;
    f16 h[8];
};

// 8 consecutive elements starting at element offset `off` of an fp16 or fp32 array, as floats
__device__ __forceinline__ void load8f(const void* base, long off, bool f32, float (&v)[8]) {
    if (f32) {
        const float* q = reinterpret_cast<const float*>(base) + off;
        const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        H8 h; h.u = ldg16(reinterpret_cast<const f16*>(base) + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)h.h[j];
    }
}
__device__ __forceinline__ void store8h(f16* p, const float (&v)[8]) {
    H8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.h[j] = (f16)v[j];
    stg16(p, o.u);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU, as torch.nn.functional.gelu(approximate="none") — /root/reference/model/attention.py:385-388
// 2 gelu(x) = x + |x| erf(|x| / sqrt 2), erf by Abramowitz-Stegun 7.1.26: erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z),
// |error| <= 1.5e-7 — 13 straight-line VALU instructions (two transcendental) where ocml's erff is ~32 with both of its branches executed
// by a diverged wave.  That matters: this runs in GEMM epilogues BESIDE the matrix pipe, where VALU time adds to MFMA time instead of
// hiding behind it (tools/probes/README.md); the fused feed-forward kernel has used the same form since it was written.
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
    const float zs = ax * 0.84932180028801904272f;          // |x| sqrt(log2(e) / 2): exp(-x^2 / 2) = exp2(-zs^2)
    const float e = __builtin_amdgcn_exp2f(-zs * zs);
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    q *= t;
    return 0.5f * fmaf(ax, fmaf(-q, e, 1.0f), x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware, bijective remap of a 1-D block id: consecutive *logical* ids land on the same XCD (block b runs on
// XCD b % 8 on MI355X), so tiles that share an operand panel share that XCD's private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, local = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}
