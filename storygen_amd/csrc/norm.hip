// GroupNorm(+SiLU) over NHWC and LayerNorm over rows, fp16 in/out, fp32 statistics (HBM-bound kernels: every access
// is a 16-byte, row-contiguous vector; statistics are deterministic — no atomics).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int GN_MAX_GROUPS = 64;
constexpr int GN_MAX_CHUNKS = 64;    // partial-sum rows per batch: every pass-2 workgroup re-reads all of them
constexpr int GNW_MAX_CHUNKS = 256;  // ... of the wide forward pair (one round of workgroups on 256 CUs); sizes the workspace

struct GnParams {
    const void* x; long ldx; int x_f32;
    f16* y; long ldy; int pad_w;          // pad_w > 0: y is [B, H+2, pad_w+2, ldy] and only its interior is written
    f16* xcopy; long ldxc;                // optional raw fp16 copy of x (MFMA operand for the 1x1 shortcut)
    const f16* gamma; const f16* beta;
    int HW, C, G, cpg, nchunks, rows_per_chunk, apply_rows;   // apply_rows: pixel rows per workgroup of pass 2
    float eps; int silu;
    float* ws;   // [B][nchunks][G][2] shifted partial sums
    // statistics from the producers' epilogues (sg_gemm_desc.stats): source i covers channels [ps_c0[i], ps_c0[i] + ps_nc[i]) of x
    // with per-(row tile of ps_rows[i] pixels, channel) sums: ps[i][((b * tiles_i + tile) * 2 + plane) * ps_nc[i] + (c - ps_c0[i])]
    const float* ps[2]; int ps_rows[2], ps_c0[2], ps_nc[2];
    // x as the unreduced output of a split-K launch (sg_groupnorm_desc.split_*), gn_fused_kernel<true> only
    const float* sp_ws; int sp_splits; long sp_MN;
    const f16* sp_bias; const float* sp_rowbias; long sp_rowbias_ld;
    const void* sp_res; long sp_ldr; int sp_res_f32;
    void* sp_out; long sp_ldo; int sp_out_f32, sp_round;
};

__device__ __forceinline__ float load1f(const void* base, long off, bool f32) {
    return f32 ? reinterpret_cast<const float*>(base)[off] : (float)reinterpret_cast<const f16*>(base)[off];
}

// Thread layout shared by both passes: TW = min(C/8, 256) threads span one pixel row's channel vectors (looping
// when C/8 > 256), 256/TW pixel rows are processed per pass.
// Pass 1: per-(batch, chunk, group) sums of (x - pivot) and (x - pivot)^2, pivot = x[b, pixel 0, first channel of
// the group] (a sample of the distribution, so E[(x-K)^2] - E[x-K]^2 has no catastrophic cancellation).
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnParams p) {
    __shared__ float s_sum[2560], s_sq[2560];   // per-channel partials of one thread-row (C <= 2560)
    const int t = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const int vpr = p.C / 8;
    const int tw = vpr < 256 ? vpr : 256, rpp = 256 / tw;
    const int my_row = t / tw, my_col = t - my_row * tw;
    const long xb = (long)b * p.HW * p.ldx;   // element offset of this batch
    const bool f32 = p.x_f32;
    const int p0 = chunk * p.rows_per_chunk, p1 = min(p.HW, p0 + p.rows_per_chunk);
    float gsum = 0.f, gsq = 0.f;   // accumulators of thread t < G (group t)
    for (int cv0 = 0; cv0 < vpr; cv0 += tw) {
        const int cv = cv0 + my_col;
        float s[8], q[8], piv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = q[j] = piv[j] = 0.f;
        const bool active = my_row < rpp && cv < vpr;
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) piv[j] = load1f(p.x, xb + ((cv * 8 + j) / p.cpg) * p.cpg, f32);
            int px = p0 + my_row;
            for (; px + 3 * rpp < p1; px += 4 * rpp) {      // 4 independent row loads in flight per thread
                float v[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) load8f(p.x, xb + (long)(px + u * rpp) * p.ldx + cv * 8, f32, v[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float d = v[u][j] - piv[j];
                        s[j] += d; q[j] += d * d;
                    }
            }
            for (; px < p1; px += rpp) {
                float v[8];
                load8f(p.x, xb + (long)px * p.ldx + cv * 8, f32, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[j] - piv[j];
                    s[j] += d; q[j] += d * d;
                }
            }
        }
        // reduce the rpp thread-rows channel-wise, then channels -> groups (sequential, deterministic)
        for (int r = 0; r < rpp; ++r) {
            __syncthreads();
            if (active && my_row == r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ch = (cv - cv0) * 8 + j;
                    if (r == 0) { s_sum[ch] = s[j]; s_sq[ch] = q[j]; }
                    else { s_sum[ch] += s[j]; s_sq[ch] += q[j]; }
                }
            }
        }
        __syncthreads();
        if (t < p.G) {
            const int c_lo = max(t * p.cpg, cv0 * 8), c_hi = min((t + 1) * p.cpg, (cv0 + tw) * 8);
            for (int ch = c_lo; ch < c_hi; ++ch) { gsum += s_sum[ch - cv0 * 8]; gsq += s_sq[ch - cv0 * 8]; }
        }
        __syncthreads();
    }
    if (t < p.G) {
        float* w = p.ws + (((long)b * p.nchunks + chunk) * p.G + t) * 2;
        w[0] = gsum; w[1] = gsq;
    }
}

// Pass 2: reduce the chunk partials, then y = silu?((x - mean) * rstd * gamma + beta).
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnParams p) {
    __shared__ float s_mean[GN_MAX_GROUPS], s_rstd[GN_MAX_GROUPS];
    const int t = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const long xb = (long)b * p.HW * p.ldx;
    const bool f32 = p.x_f32;
    const int pw = p.pad_w;
    // padded output: pixel (yy, xx) of image b lives at row ((b*(H+2) + yy+1)*(pw+2) + xx+1)
    f16* yb = pw ? p.y + ((long)b * (p.HW / pw + 2) * (pw + 2)) * p.ldy : p.y + (long)b * p.HW * p.ldy;
    f16* cb = p.xcopy ? p.xcopy + (long)b * p.HW * p.ldxc : nullptr;
    {   // reduce the chunk partials: thread (part, group) sums every `parts`-th chunk, then a fixed-order LDS tree
        __shared__ float s_ps[256], s_pq[256];
        const int parts = 256 / p.G;                 // G <= 64 -> parts >= 4
        const int grp = t % p.G, part = t / p.G;
        float s = 0.f, q = 0.f;
        if (part < parts)
            for (int c = part; c < p.nchunks; c += parts) {
                const float* w = p.ws + (((long)b * p.nchunks + c) * p.G + grp) * 2;
                s += w[0]; q += w[1];
            }
        s_ps[t] = s; s_pq[t] = q;
        __syncthreads();
        if (t < p.G) {
            s = 0.f; q = 0.f;
            for (int k = 0; k < parts; ++k) { s += s_ps[k * p.G + t]; q += s_pq[k * p.G + t]; }
            const float n = (float)p.HW * (float)p.cpg;
            const float piv = load1f(p.x, xb + t * p.cpg, f32);
            const float md = s / n;                       // E[x - K]
            const float var = fmaxf(q / n - md * md, 0.f);
            s_mean[t] = piv + md;
            s_rstd[t] = rsqrtf(var + p.eps);
        }
    }
    __syncthreads();
    const int vpr = p.C / 8;
    const int tw = vpr < 256 ? vpr : 256, rpp = 256 / tw;
    const int my_row = t / tw, my_col = t - my_row * tw;
    if (my_row >= rpp) return;
    const int p0 = chunk * p.apply_rows, p1 = min(p.HW, p0 + p.apply_rows);
    for (int cv = my_col; cv < vpr; cv += tw) {
        float sc[8], sh[8];
        H8 g, be; g.u = ldg16(p.gamma + cv * 8); be.u = ldg16(p.beta + cv * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int grp = (cv * 8 + j) / p.cpg;
            sc[j] = s_rstd[grp] * (float)g.h[j];
            sh[j] = (float)be.h[j] - s_mean[grp] * sc[j];
        }
        auto emit = [&](int px, const float (&v)[8]) {
            float o[8];
            if (cb) store8h(cb + (long)px * p.ldxc + cv * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float r = v[j] * sc[j] + sh[j];
                o[j] = p.silu ? silu_f(r) : r;
            }
            long orow = px;
            if (pw) { const int yy = px / pw, xx = px - yy * pw; orow = (long)(yy + 1) * (pw + 2) + xx + 1; }
            store8h(yb + orow * p.ldy + cv * 8, o);
        };
        int px = p0 + my_row;
        for (; px + 3 * rpp < p1; px += 4 * rpp) {        // 4 independent rows in flight per thread
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) load8f(p.x, xb + (long)(px + u * rpp) * p.ldx + cv * 8, f32, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) emit(px + u * rpp, v[u]);
        }
        for (; px < p1; px += rpp) {
            float v[8];
            load8f(p.x, xb + (long)px * p.ldx + cv * 8, f32, v);
            emit(px, v);
        }
    }
}

// ---- wide two-pass variant (large feature maps) ------------------------------------------------------------------
// 1024-thread workgroups, thread (row r, column cv) owns the 8-channel vector cv of every rpp-th pixel row of its
// chunk: one CU keeps 16 waves x 4 rows x 32 B of loads in flight, which is what it takes to pull MALL/HBM bandwidth
// with <= 64 chunks per batch (the partial-sum table every pass-2 workgroup re-reads stays 16 KB).  Both passes use the
// SAME (block id -> pixel rows) mapping: block b runs on XCD b % 8, so pass 2 re-reads its slab from the XCD-private
// L2 that pass 1 pulled it into.  Needs cpg >= 8 (a vector touches at most two groups) and C <= 2560.
constexpr int GNW_NT = 1024;
constexpr int GNW_MAX_VPR = 320;

__global__ __launch_bounds__(GNW_NT) void gn_stats_wide_kernel(const GnParams p) {
    __shared__ float4 s_part[GNW_NT];        // per thread: (sum, sq) of its vector's low group, (sum, sq) of its high group
    __shared__ float4 s_col[GNW_MAX_VPR];    // the same per vector column, after the fixed-order reduction over thread rows
    const int t = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const int vpr = p.C >> 3, rpp = GNW_NT / vpr;
    const int my_row = t / vpr, cv = t - my_row * vpr;
    const long xb = (long)b * p.HW * p.ldx;
    const bool f32 = p.x_f32;
    const int p0 = chunk * p.rows_per_chunk, p1 = min(p.HW, p0 + p.rows_per_chunk);
    const int c0 = cv * 8;
    const int g_lo = c0 / p.cpg, g_hi = (c0 + 7) / p.cpg;
    const int nlo = min(8, (g_lo + 1) * p.cpg - c0);      // the vector's first nlo channels belong to g_lo
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (my_row < rpp) {
        // pivot = x[b, pixel 0, first channel of the group] (see gn_stats_kernel)
        const float piv_lo = load1f(p.x, xb + g_lo * p.cpg, f32), piv_hi = load1f(p.x, xb + g_hi * p.cpg, f32);
        float piv[8], s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { piv[j] = j < nlo ? piv_lo : piv_hi; s[j] = q[j] = 0.f; }
        for (int base = p0 + my_row; base < p1; base += 4 * rpp) {      // up to 4 independent row loads in flight
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = base + u * rpp;
                if (px < p1) load8f(p.x, xb + (long)px * p.ldx + c0, f32, v[u]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[u][j] = piv[j];       // contributes exactly zero
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float d = v[u][j] - piv[j];
                    s[j] += d; q[j] += d * d;
                }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < nlo) { acc.x += s[j]; acc.y += q[j]; }
            else { acc.z += s[j]; acc.w += q[j]; }
        }
    }
    s_part[t] = acc;
    __syncthreads();
    if (t < vpr) {
        float4 a = s_part[t];
        for (int r = 1; r < rpp; ++r) {
            const float4 o = s_part[r * vpr + t];
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        s_col[t] = a;
    }
    __syncthreads();
    if (t < p.G) {
        const int cv_a = (t * p.cpg) >> 3, cv_b = ((t + 1) * p.cpg - 1) >> 3;
        float gs = 0.f, gq = 0.f;
        for (int c = cv_a; c <= cv_b; ++c) {
            const float4 a = s_col[c];
            const int lo = (c * 8) / p.cpg, hi = (c * 8 + 7) / p.cpg;
            if (lo == t) { gs += a.x; gq += a.y; }
            if (hi == t && hi != lo) { gs += a.z; gq += a.w; }
        }
        float* w = p.ws + (((long)b * p.nchunks + chunk) * p.G + t) * 2;
        w[0] = gs; w[1] = gq;
    }
}

__global__ __launch_bounds__(GNW_NT) void gn_apply_wide_kernel(const GnParams p) {
    __shared__ float s_mean[GN_MAX_GROUPS], s_rstd[GN_MAX_GROUPS];
    __shared__ float s_ps[GNW_NT], s_pq[GNW_NT];
    __shared__ float s_cs[GNW_MAX_VPR * 8], s_cq[GNW_MAX_VPR * 8], s_cn[GNW_MAX_VPR * 8];   // per-(slot, channel) partials of the coalesced merge
    const int t = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const int vpr = p.C >> 3, rpp = GNW_NT / vpr;
    const int my_row = t / vpr, cv = t - my_row * vpr;
    const bool active = my_row < rpp;
    const long xb = (long)b * p.HW * p.ldx;
    const bool f32 = p.x_f32;
    const int pw = p.pad_w;
    const int p0 = chunk * p.rows_per_chunk, p1 = min(p.HW, p0 + p.rows_per_chunk);
    const int c0 = cv * 8;
    float v[4][8];
    int base = p0 + my_row;
    auto load_batch = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int px = base + u * rpp;
            if (active && px < p1) load8f(p.x, xb + (long)px * p.ldx + c0, f32, v[u]);
        }
    };
    load_batch();       // the first rows travel while the statistics are being reduced
    H8 gam, bet;        // ... and so do the affine parameters (requested after the statistics they would cost a second round trip)
    gam.u = bet.u = make_uint4(0, 0, 0, 0);
    if (active) { gam.u = ldg16(p.gamma + c0); bet.u = ldg16(p.beta + c0); }
    if (p.ps[0]) {
        // Statistics from the producers' partials (fused epilogues and split-K second passes), channel-coalesced (round 4).  The partials
        // of a sample are C x tiles x 8 bytes and EVERY workgroup of the sample needs all of them; the round-3 merge walked them
        // group-major — lanes 40 bytes apart, ~20 cache lines per wave load, ~10 000 cycles of the CU's texture path per workgroup, as much
        // as the rows it normalises.  Here a thread owns ONE channel (and, while C <= 512, every slots-th row tile of it): consecutive
        // lanes read consecutive floats of one partial row, 2 lines per wave load.  Two-level Chan merge (one pivot per GROUP cost up to 3 of fp32's 7 digits when its first entry was an outlier):
        //   thread (slot, c): over its tiles, about the pivot K = the mean of its first tile:  n_t, s_t, M2_t   (one channel's tiles only:
        //       the cancellation (s_t / n_t - K)^2 is between row tiles of one channel)
        //   group: mean = sum s_t / N,  M2 = sum [ M2_t + n_t (s_t / n_t - mean)^2 ]  over its cpg x slots thread partials (non-negative terms).
        const int C = p.C;
        const int slots = C <= GNW_NT ? GNW_NT / C : 1;
        const int t0 = p.HW / p.ps_rows[0], t1 = p.ps[1] ? p.HW / p.ps_rows[1] : 0;
        constexpr int GNM_U = 8;
        for (int cb = 0; cb < C; cb += GNW_NT) {
            const int slot = C <= GNW_NT ? t / C : 0;
            const int c = C <= GNW_NT ? t - slot * C : cb + t;
            const bool owner = slot < slots && c < C;
            const int cc = owner ? c : 0;
            const int src = (p.ps[1] && cc >= p.ps_c0[1]) ? 1 : 0;
            const int tiles = src ? t1 : t0;
            const int nc = p.ps_nc[src];
            const float rows = (float)p.ps_rows[src], inv_rows = 1.0f / rows;
            const float* base = p.ps[src] + (long)b * tiles * 2 * nc + (cc - p.ps_c0[src]);
            float a_s = 0.f, a_q = 0.f, a_n = 0.f, K = 0.f;
            bool have_k = false;
            for (int i0 = owner ? slot : tiles; i0 < tiles; i0 += GNM_U * slots) {
                float s1[GNM_U], s2[GNM_U];
#pragma unroll
                for (int u = 0; u < GNM_U; ++u) {
                    const int tl = i0 + u * slots;
                    const float* e = base + (long)(tl < tiles ? tl : i0) * 2 * nc;      // clamped: no predicate on the loads
                    s1[u] = e[0]; s2[u] = e[nc];
                }
#pragma unroll
                for (int u = 0; u < GNM_U; ++u)
                    if (owner && i0 + u * slots < tiles) {
                        const float me = s1[u] * inv_rows;
                        if (!have_k) { K = me; have_k = true; }
                        const float dm = me - K;
                        a_s += s1[u];
                        a_n += rows;
                        a_q += fmaxf(s2[u] - s1[u] * me, 0.f) + rows * dm * dm;
                    }
            }
            if (owner) {
                const float mt = a_n > 0.f ? a_s / a_n : 0.f, dk = mt - K;
                const int idx = slot * C + c;
                s_cs[idx] = a_s; s_cq[idx] = fmaxf(a_q - a_n * dk * dk, 0.f); s_cn[idx] = a_n;
            }
        }
        __syncthreads();
        // group g = half-wave (16 waves x 2 halves = 32 groups per round): its cpg x slots partials are summed by 32 lanes with xor
        // shuffles (offsets < 32 stay inside the half), mean first, then the non-negative M2 terms about it
        {
            const int lane = t & 63, l32 = lane & 31;
            const float ntot = (float)p.HW * (float)p.cpg;
            for (int g = (t >> 6) * 2 + (lane >> 5); g < p.G; g += 2 * (GNW_NT / 64)) {
                const int c_lo = g * p.cpg;
                float sm = 0.f;
                for (int sl = 0; sl < slots; ++sl)
                    for (int k = l32; k < p.cpg; k += 32) sm += s_cs[sl * C + c_lo + k];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
                const float mean = sm / ntot;
                float m2 = 0.f;
                for (int sl = 0; sl < slots; ++sl)
                    for (int k = l32; k < p.cpg; k += 32) {
                        const int idx = sl * C + c_lo + k;
                        const float nt = s_cn[idx];
                        const float d = nt > 0.f ? s_cs[idx] / nt - mean : 0.f;
                        m2 += s_cq[idx] + nt * d * d;
                    }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) m2 += __shfl_xor(m2, o);
                if (l32 == 0) {
                    s_mean[g] = mean;
                    s_rstd[g] = rsqrtf(m2 / ntot + p.eps);
                }
            }
        }
        __syncthreads();
    } else
    {   // reduce the chunk partials: thread (part, group) sums every `parts`-th chunk, then a fixed-order LDS pass
        const int parts = GNW_NT / p.G;
        const int grp = t % p.G, part = t / p.G;
        float s = 0.f, q = 0.f;
        if (part < parts)
            for (int c = part; c < p.nchunks; c += parts) {
                const float* w = p.ws + (((long)b * p.nchunks + c) * p.G + grp) * 2;
                s += w[0]; q += w[1];
            }
        s_ps[t] = s; s_pq[t] = q;
        __syncthreads();
        if (t < p.G) {
            s = 0.f; q = 0.f;
            for (int k = 0; k < parts; ++k) { s += s_ps[k * p.G + t]; q += s_pq[k * p.G + t]; }
            const float n = (float)p.HW * (float)p.cpg;
            const float piv = load1f(p.x, xb + t * p.cpg, f32);
            const float md = s / n;
            const float var = fmaxf(q / n - md * md, 0.f);
            s_mean[t] = piv + md;
            s_rstd[t] = rsqrtf(var + p.eps);
        }
        __syncthreads();
    }
    if (!active) return;
    f16* yb = pw ? p.y + ((long)b * (p.HW / pw + 2) * (pw + 2)) * p.ldy : p.y + (long)b * p.HW * p.ldy;
    f16* cb = p.xcopy ? p.xcopy + (long)b * p.HW * p.ldxc : nullptr;
    float sc[8], sh[8];
    const float inv_cpg = 1.0f / (float)p.cpg, inv_pw = pw ? 1.0f / (float)pw : 0.f;    // exact quotients for operands < 2^20
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int grp = (int)(((float)(c0 + j) + 0.5f) * inv_cpg);
        sc[j] = s_rstd[grp] * (float)gam.h[j];
        sh[j] = (float)bet.h[j] - s_mean[grp] * sc[j];
    }
    while (base < p1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int px = base + u * rpp;
            if (px < p1) {
                float o[8];
                if (cb) store8h(cb + (long)px * p.ldxc + c0, v[u]);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float r = v[u][j] * sc[j] + sh[j];
                    o[j] = p.silu ? silu_f(r) : r;
                }
                long orow = px;
                if (pw) { const int yy = (int)(((float)px + 0.5f) * inv_pw), xx = px - yy * pw; orow = (long)(yy + 1) * (pw + 2) + xx + 1; }
                store8h(yb + orow * p.ldy + c0, o);
            }
        }
        base += 4 * rpp;
        load_batch();
    }
}

// ---- GroupNorm(+SiLU) backward (training step, BASELINE config 4; formulas: oracle/storygen_backward.py) ----------------
// Validated on MI355X in round 2 (tests/test_backward_gpu.py::test_groupnorm_bwd).
// y = act(xhat * gamma + beta), act = SiLU or identity.  With g = dact * gamma (dact = dy * act'(n), n = xhat*gamma+beta):
//   dx = rstd * (g - mean_group(g) - xhat * mean_group(g * xhat)).
// Three launches with the wide geometry: gn_stats_wide_kernel (statistics of x, exactly as in the forward pass), then
// pass A below (per-chunk sums of g and g*xhat) and pass B (dx, optionally + a residual gradient, written as the fp32
// stream or as the zero-bordered fp16 input of the next dgrad convolution).
struct GnBwdParams {
    GnParams f;                 // x, gamma, beta, geometry, eps, silu, ws (partials of x) as in the forward pass
    const void* dy; long lddy; int dy_f32;
    float* ws2;                 // [B][nchunks][G][2] partial sums of (g, g * xhat)
    const float* res; long ldr; // optional residual gradient added to dx (fp32 output only)
    float* out32; long ldo32;   // fp32 output [B, HW, C] ...
    f16* out16; long ldo16; int pad_w;   // ... or fp16 output (pad_w > 0: interior of [B, H+2, pad_w+2, C])
};

// mean / rstd of every group of batch b from the forward statistics partials (same code as gn_apply_wide_kernel's prologue)
__device__ __forceinline__ void gnw_group_stats(const GnParams& p, int b, long xb, float* s_mean, float* s_rstd, float* s_ps, float* s_pq) {
    const int t = threadIdx.x;
    const int parts = GNW_NT / p.G;
    const int grp = t % p.G, part = t / p.G;
    float s = 0.f, q = 0.f;
    if (part < parts)
        for (int c = part; c < p.nchunks; c += parts) {
            const float* w = p.ws + (((long)b * p.nchunks + c) * p.G + grp) * 2;
            s += w[0]; q += w[1];
        }
    s_ps[t] = s; s_pq[t] = q;
    __syncthreads();
    if (t < p.G) {
        s = 0.f; q = 0.f;
        for (int k = 0; k < parts; ++k) { s += s_ps[k * p.G + t]; q += s_pq[k * p.G + t]; }
        const float n = (float)p.HW * (float)p.cpg;
        const float piv = load1f(p.x, xb + t * p.cpg, p.x_f32);
        const float md = s / n;
        const float var = fmaxf(q / n - md * md, 0.f);
        s_mean[t] = piv + md;
        s_rstd[t] = rsqrtf(var + p.eps);
    }
    __syncthreads();
}

// g[j] and xhat[j] of one 8-channel vector of one pixel
__device__ __forceinline__ void gnb_g_xhat(const float (&x)[8], const float (&dy)[8], const float (&mu)[8], const float (&rs)[8],
                                           const float (&gm)[8], const float (&bt)[8], bool silu, float (&g)[8], float (&xh)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xh[j] = (x[j] - mu[j]) * rs[j];
        float d = dy[j];
        if (silu) {
            const float n = xh[j] * gm[j] + bt[j];
            const float sg = 1.0f / (1.0f + __expf(-n));
            d *= sg * (1.0f + n * (1.0f - sg));
        }
        g[j] = d * gm[j];
    }
}

__global__ __launch_bounds__(GNW_NT) void gn_bwd_sums_kernel(const GnBwdParams q) {
    __shared__ float s_mean[GN_MAX_GROUPS], s_rstd[GN_MAX_GROUPS];
    __shared__ float s_ps[GNW_NT], s_pq[GNW_NT];
    __shared__ float4 s_col[GNW_MAX_VPR];
    const GnParams& p = q.f;
    const int t = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const int vpr = p.C >> 3, rpp = GNW_NT / vpr;
    const int my_row = t / vpr, cv = t - my_row * vpr;
    const long xb = (long)b * p.HW * p.ldx, db = (long)b * p.HW * q.lddy;
    gnw_group_stats(p, b, xb, s_mean, s_rstd, s_ps, s_pq);
    const int p0 = chunk * p.rows_per_chunk, p1 = min(p.HW, p0 + p.rows_per_chunk);
    const int c0 = cv * 8;
    const int g_lo = c0 / p.cpg;
    const int nlo = min(8, (g_lo + 1) * p.cpg - c0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);      // (sum g, sum g*xhat) of the vector's low group, of its high group
    if (my_row < rpp) {
        float mu[8], rs[8], gm[8], bt[8];
        H8 gv, bv; gv.u = ldg16(p.gamma + c0); bv.u = ldg16(p.beta + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int grp = (c0 + j) / p.cpg;
            mu[j] = s_mean[grp]; rs[j] = s_rstd[grp]; gm[j] = (float)gv.h[j]; bt[j] = (float)bv.h[j];
        }
        float sg[8], sx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sg[j] = sx[j] = 0.f;
        for (int px = p0 + my_row; px < p1; px += rpp) {
            float x[8], dy[8], g[8], xh[8];
            load8f(p.x, xb + (long)px * p.ldx + c0, p.x_f32, x);
            load8f(q.dy, db + (long)px * q.lddy + c0, q.dy_f32, dy);
            gnb_g_xhat(x, dy, mu, rs, gm, bt, p.silu != 0, g, xh);
#pragma unroll
            for (int j = 0; j < 8; ++j) { sg[j] += g[j]; sx[j] += g[j] * xh[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < nlo) { acc.x += sg[j]; acc.y += sx[j]; }
            else { acc.z += sg[j]; acc.w += sx[j]; }
        }
    }
    // fixed-order reduction over the thread rows, then vector columns -> groups (as in gn_stats_wide_kernel)
    __shared__ float4 s_part[GNW_NT];
    s_part[t] = acc;
    __syncthreads();
    if (t < vpr) {
        float4 a = s_part[t];
        for (int r = 1; r < rpp; ++r) {
            const float4 o = s_part[r * vpr + t];
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
        s_col[t] = a;
    }
    __syncthreads();
    if (t < p.G) {
        const int cv_a = (t * p.cpg) >> 3, cv_b = ((t + 1) * p.cpg - 1) >> 3;
        float gs = 0.f, gq = 0.f;
        for (int c = cv_a; c <= cv_b; ++c) {
            const float4 a = s_col[c];
            const int lo = (c * 8) / p.cpg, hi = (c * 8 + 7) / p.cpg;
            if (lo == t) { gs += a.x; gq += a.y; }
            if (hi == t && hi != lo) { gs += a.z; gq += a.w; }
        }
        float* w = q.ws2 + (((long)b * p.nchunks + chunk) * p.G + t) * 2;
        w[0] = gs; w[1] = gq;
    }
}

__global__ __launch_bounds__(GNW_NT) void gn_bwd_apply_kernel(const GnBwdParams q) {
    __shared__ float s_mean[GN_MAX_GROUPS], s_rstd[GN_MAX_GROUPS], s_m1[GN_MAX_GROUPS], s_m2[GN_MAX_GROUPS];
    __shared__ float s_ps[GNW_NT], s_pq[GNW_NT];
    const GnParams& p = q.f;
    const int t = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const int vpr = p.C >> 3, rpp = GNW_NT / vpr;
    const int my_row = t / vpr, cv = t - my_row * vpr;
    const long xb = (long)b * p.HW * p.ldx, db = (long)b * p.HW * q.lddy;
    gnw_group_stats(p, b, xb, s_mean, s_rstd, s_ps, s_pq);
    {   // group means of g and g * xhat from pass A's partials (same fixed-order scheme)
        const int parts = GNW_NT / p.G;
        const int grp = t % p.G, part = t / p.G;
        float s = 0.f, u = 0.f;
        if (part < parts)
            for (int c = part; c < p.nchunks; c += parts) {
                const float* w = q.ws2 + (((long)b * p.nchunks + c) * p.G + grp) * 2;
                s += w[0]; u += w[1];
            }
        s_ps[t] = s; s_pq[t] = u;
        __syncthreads();
        if (t < p.G) {
            s = 0.f; u = 0.f;
            for (int k = 0; k < parts; ++k) { s += s_ps[k * p.G + t]; u += s_pq[k * p.G + t]; }
            const float n = (float)p.HW * (float)p.cpg;
            s_m1[t] = s / n; s_m2[t] = u / n;
        }
        __syncthreads();
    }
    if (my_row >= rpp) return;
    const int p0 = chunk * p.rows_per_chunk, p1 = min(p.HW, p0 + p.rows_per_chunk);
    const int c0 = cv * 8;
    float mu[8], rs[8], gm[8], bt[8], m1[8], m2[8];
    {
        H8 gv, bv; gv.u = ldg16(p.gamma + c0); bv.u = ldg16(p.beta + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int grp = (c0 + j) / p.cpg;
            mu[j] = s_mean[grp]; rs[j] = s_rstd[grp]; m1[j] = s_m1[grp]; m2[j] = s_m2[grp];
            gm[j] = (float)gv.h[j]; bt[j] = (float)bv.h[j];
        }
    }
    const int pw = q.pad_w;
    for (int px = p0 + my_row; px < p1; px += rpp) {
        float x[8], dy[8], g[8], xh[8], o[8];
        load8f(p.x, xb + (long)px * p.ldx + c0, p.x_f32, x);
        load8f(q.dy, db + (long)px * q.lddy + c0, q.dy_f32, dy);
        gnb_g_xhat(x, dy, mu, rs, gm, bt, p.silu != 0, g, xh);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs[j] * (g[j] - m1[j] - xh[j] * m2[j]);
        if (q.out32) {
            if (q.res) {
                const float* r = q.res + ((long)b * p.HW + px) * q.ldr + c0;
                const float4 a = *reinterpret_cast<const float4*>(r), c = *reinterpret_cast<const float4*>(r + 4);
                o[0] += a.x; o[1] += a.y; o[2] += a.z; o[3] += a.w; o[4] += c.x; o[5] += c.y; o[6] += c.z; o[7] += c.w;
            }
            float* w = q.out32 + ((long)b * p.HW + px) * q.ldo32 + c0;
            *reinterpret_cast<float4*>(w) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(w + 4) = make_float4(o[4], o[5], o[6], o[7]);
        } else {
            long orow = (long)b * p.HW + px;
            if (pw) {
                const int yy = px / pw, xx = px - yy * pw;
                orow = ((long)b * (p.HW / pw + 2) + yy + 1) * (pw + 2) + xx + 1;
            }
            store8h(q.out16 + orow * q.ldo16 + c0, o);
        }
    }
}

// Small feature maps (16x16 / 8x8 latent levels): one workgroup owns one (batch, group) slab, keeps it in registers,
// computes exact two-pass statistics and writes the normalised output — one launch, one read of x.
constexpr int GNF_MAXI = 24;   // 4-element items per thread: slabs of up to 256 * 24 * 4 = 24576 values
// measured (tools/bench_norm.py, MI355X): the one-launch kernel wins up to 10 240-element slabs (8x8 level, 16x16 up to 1280
// channels: 5.7-10.9 us vs 10-11 us for the two launches of the wide pair) and loses above (16x16x2560: 18.3 vs 12.9 us,
// 32x32x640: 22.7 vs 11.9 us — 40-byte row pieces, 128 workgroups)
constexpr long GNF_DEFAULT_MAX = 10240;

// SPLIT (round 4): x is the unreduced output of a split-K convolution — the slab's values are the sums of the fp32 partial slices
// (slice order) + bias + temb row + residual, i.e. exactly what splitk_reduce_kernel would have stored, computed while the slab is
// loaded; the reduced tensor is written only if somebody else needs it.  conv1 -> norm2 and conv2 -> Transformer2DModel.norm at the
// 16x16 / 8x8 levels lose their second pass (and the round trip of the reduced tensor through memory).
// NT threads per workgroup, MAXI items per thread: 256 x 24 for large slabs; 1024 x 4 (slabs up to 16 384 values) keeps four times
// as many loads in flight per CU — one workgroup per (batch, group) is at most 128 workgroups, so each CU's memory-level parallelism is
// what bounds the kernel, above all when the slab arrives as S split-K slices.
template <bool SPLIT, int NT, int MAXI>
__global__ __launch_bounds__(NT) void gn_fused_kernel(const GnParams p) {
    static_assert(MAXI % 4 == 0, "the split pre-pass works in batches of four items");
    __shared__ float s_red[NT / 64];
    const int t = threadIdx.x, b = blockIdx.y, g = blockIdx.x;
    const int q = p.cpg / 4;                    // 4-channel items per pixel in this group
    const int items = p.HW * q;
    const long xb = (long)b * p.HW * p.ldx + (long)g * p.cpg;
    const bool f32 = p.x_f32;
    float v[MAXI][4];
    f16x4 gm[MAXI], bt[MAXI];    // the affine parameters travel with the data, not after the statistics
    if constexpr (SPLIT) {
        // Batched pre-pass: 4 items x 4 slices = 16 independent 16-byte loads in flight per thread, addresses clamped instead of
        // predicated (a load inside a per-item `if` is waited for where it is issued: one exposed round trip per item and slice group).
        // Slices are added in slice order, then bias, temb row, residual — the order of splitk_reduce_kernel: bit-identical values.
#pragma unroll
        for (int i0 = 0; i0 < MAXI; i0 += 4) {
            if (i0 * NT < items) {                         // block-uniform
                const float* sl[4];
                long mm[4];
                int nn[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int it = min(t + (i0 + k) * NT, items - 1);
                    const int px = it / q, c4 = it - px * q;
                    mm[k] = (long)b * p.HW + px;
                    nn[k] = g * p.cpg + c4 * 4;
                    sl[k] = p.sp_ws + mm[k] * p.C + nn[k];
                }
                float4 a[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int z0 = 0; z0 < p.sp_splits; z0 += 4) {
                    float4 q4[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const long zo = (long)min(z0 + u, p.sp_splits - 1) * p.sp_MN;
#pragma unroll
                        for (int k = 0; k < 4; ++k) q4[k][u] = *reinterpret_cast<const float4*>(sl[k] + zo);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (z0 + u < p.sp_splits) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) { a[k].x += q4[k][u].x; a[k].y += q4[k][u].y; a[k].z += q4[k][u].z; a[k].w += q4[k][u].w; }
                        }
                }
                if (p.sp_bias) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f16x4 bb = *reinterpret_cast<const f16x4*>(p.sp_bias + nn[k]);
                        a[k].x += (float)bb[0]; a[k].y += (float)bb[1]; a[k].z += (float)bb[2]; a[k].w += (float)bb[3];
                    }
                }
                if (p.sp_rowbias) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 rb = *reinterpret_cast<const float4*>(p.sp_rowbias + (long)b * p.sp_rowbias_ld + nn[k]);
                        a[k].x += rb.x; a[k].y += rb.y; a[k].z += rb.z; a[k].w += rb.w;
                    }
                }
                if (p.sp_res) {
                    if (p.sp_res_f32) {
                        float4 r4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) r4[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.sp_res) + mm[k] * p.sp_ldr + nn[k]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) { a[k].x += r4[k].x; a[k].y += r4[k].y; a[k].z += r4[k].z; a[k].w += r4[k].w; }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f16x4 r4 = *reinterpret_cast<const f16x4*>(reinterpret_cast<const f16*>(p.sp_res) + mm[k] * p.sp_ldr + nn[k]);
                            a[k].x += (float)r4[0]; a[k].y += (float)r4[1]; a[k].z += (float)r4[2]; a[k].w += (float)r4[3];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool mine = t + (i0 + k) * NT < items;
                    if (mine && p.sp_out && p.sp_out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.sp_out) + mm[k] * p.sp_ldo + nn[k]) = a[k];
                    if (p.sp_round) {      // the tensor is (or would have been) stored in fp16: normalise what a reader of it would see
                        const f16x4 hh = {(f16)a[k].x, (f16)a[k].y, (f16)a[k].z, (f16)a[k].w};
                        if (mine && p.sp_out && !p.sp_out_f32) *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.sp_out) + mm[k] * p.sp_ldo + nn[k]) = hh;
                        a[k] = make_float4((float)hh[0], (float)hh[1], (float)hh[2], (float)hh[3]);
                    }
                    v[i0 + k][0] = mine ? a[k].x : 0.f; v[i0 + k][1] = mine ? a[k].y : 0.f;
                    v[i0 + k][2] = mine ? a[k].z : 0.f; v[i0 + k][3] = mine ? a[k].w : 0.f;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[i0 + k][0] = v[i0 + k][1] = v[i0 + k][2] = v[i0 + k][3] = 0.f;
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int it = t + i * NT;
        if constexpr (!SPLIT) v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
        gm[i] = bt[i] = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        if (it < items) {
            const int px = it / q, c4 = it - px * q;
            const long off = xb + (long)px * p.ldx + c4 * 4;
            gm[i] = *reinterpret_cast<const f16x4*>(p.gamma + g * p.cpg + c4 * 4);
            bt[i] = *reinterpret_cast<const f16x4*>(p.beta + g * p.cpg + c4 * 4);
            if constexpr (SPLIT) {
                // (v[i] was filled by the batched pre-pass above)
            } else if (f32) {
                const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.x) + off);
                v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w;
            } else {
                const f16x4 a = *reinterpret_cast<const f16x4*>(reinterpret_cast<const f16*>(p.x) + off);
                v[i][0] = (float)a[0]; v[i][1] = (float)a[1]; v[i][2] = (float)a[2]; v[i][3] = (float)a[3];
            }
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    auto block_sum = [&](float x) {
        x = wave_sum(x);
        __syncthreads();
        if ((t & 63) == 0) s_red[t >> 6] = x;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot += s_red[w];       // fixed order: deterministic
        return tot;
    };
    const float n = (float)items * 4.f;
    const float mean = block_sum(sum) / n;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
        if (t + i * NT < items) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(block_sum(sq) / n + p.eps);
    const int pw = p.pad_w;
    f16* yb = pw ? p.y + ((long)b * (p.HW / pw + 2) * (pw + 2)) * p.ldy : p.y + (long)b * p.HW * p.ldy;
    f16* cb = p.xcopy ? p.xcopy + (long)b * p.HW * p.ldxc : nullptr;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int it = t + i * NT;
        if (it < items) {
            const int px = it / q, c4 = it - px * q;
            const int ch = g * p.cpg + c4 * 4;
            f16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float r = (v[i][j] - mean) * rstd * (float)gm[i][j] + (float)bt[i][j];
                o[j] = (f16)(p.silu ? silu_f(r) : r);
            }
            long orow = px;
            if (pw) { const int yy = px / pw, xx = px - yy * pw; orow = (long)(yy + 1) * (pw + 2) + xx + 1; }
            *reinterpret_cast<f16x4*>(yb + orow * p.ldy + ch) = o;
            if (cb) {
                f16x4 c = {(f16)v[i][0], (f16)v[i][1], (f16)v[i][2], (f16)v[i][3]};
                *reinterpret_cast<f16x4*>(cb + (long)px * p.ldxc + ch) = c;
            }
        }
    }
}

struct LnParams {
    const void* x; long ldx; int x_f32; int M, C; float eps;
    const f16* g1; const f16* b1; f16* y1; long ldy1;
    const f16* g2; const f16* b2; f16* y2; long ldy2;
};

// One wave per row; the row (C <= 64*8*NV halves) stays in registers; exact two-pass mean / variance.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnParams p) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int vpr = p.C / 8;
    const long xr = (long)row * p.ldx;
    float v[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cv = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        if (cv < vpr) {
            load8f(p.x, xr + cv * 8, p.x_f32, v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i][j];
        }
    }
    const float mean = wave_sum(sum) / (float)p.C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < vpr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.C + p.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cv = lane + 64 * i;
        if (cv < vpr) {
            H8 g, be, o;
            g.u = ldg16(p.g1 + cv * 8); be.u = ldg16(p.b1 + cv * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o.h[j] = (f16)((v[i][j] - mean) * rstd * (float)g.h[j] + (float)be.h[j]);
            stg16(p.y1 + (long)row * p.ldy1 + cv * 8, o.u);
            if (p.y2) {
                g.u = ldg16(p.g2 + cv * 8); be.u = ldg16(p.b2 + cv * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) o.h[j] = (f16)((v[i][j] - mean) * rstd * (float)g.h[j] + (float)be.h[j]);
                stg16(p.y2 + (long)row * p.ldy2 + cv * 8, o.u);
            }
        }
    }
}

// Both passes are pure streaming.  Pass 1 wants enough workgroups to pull HBM (>= 1 per CU) but few partial-sum rows
// (pass 2 re-reads them all in every workgroup); pass 2 wants ~4 workgroups per CU with >= 32 KB of rows each, so that
// the fixed prologue (partials, gamma/beta) is amortised.
void gn_geometry(int B, int HW, int C, bool f32, int* nchunks, int* rows_per_chunk, int* apply_blocks, int* apply_rows) {
    const int vpr = C / 8, tw = vpr < 256 ? vpr : 256, rpp = 256 / tw;
    int n = sg_cdiv(HW, rpp * 8);                 // >= 8 passes of rows per workgroup
    const int want1 = sg_cdiv(512, B);
    if (n > want1) n = want1;
    if (n > GN_MAX_CHUNKS) n = GN_MAX_CHUNKS;
    if (n < 1) n = 1;
    *rows_per_chunk = sg_cdiv(HW, n);
    *nchunks = sg_cdiv(HW, *rows_per_chunk);
    const long row_bytes = (long)C * (f32 ? 4 : 2);
    int min_rows = (int)((32768 + row_bytes - 1) / row_bytes);
    if (min_rows < 2 * rpp) min_rows = 2 * rpp;
    int a = sg_cdiv(HW, min_rows);
    const int want2 = sg_cdiv(1024, B);
    if (a > want2) a = want2;
    if (a < 1) a = 1;
    *apply_rows = sg_cdiv(HW, a);
    *apply_blocks = sg_cdiv(HW, *apply_rows);
}

// does a GroupNorm of this shape run as the one-launch kernel?
bool gn_takes_fused(int HW, int C, int groups) {
    const SgOptions& opt = sg_options();
    const long fused_max = opt.gn_fused_max >= 0 ? opt.gn_fused_max : GNF_DEFAULT_MAX;
    const int cpg = C / groups;
    const long slab = (long)HW * cpg;
    return cpg % 4 == 0 && slab <= 256L * 4 * GNF_MAXI && slab <= fused_max && !opt.gn_no_fused;
}

}  // namespace

extern "C" size_t sg_groupnorm_workspace_bytes(int32_t B, int32_t groups) {
    return (size_t)B * GNW_MAX_CHUNKS * (size_t)groups * 2 * sizeof(float);
}

extern "C" int sg_groupnorm_uses_pstats(int32_t HW, int32_t C, int32_t groups) {
    if (HW <= 0 || C <= 0 || groups <= 0 || C % groups) return 0;
    const SgOptions& opt = sg_options();
    const int cpg = C / groups;
    if (gn_takes_fused(HW, C, groups)) return 0;      // gn_fused_kernel
    return (opt.gn_wide != 0 && cpg >= 8 && C / 8 <= GNW_MAX_VPR) ? 1 : 0;
}

extern "C" int sg_groupnorm_is_fused(int32_t HW, int32_t C, int32_t groups) {
    if (HW <= 0 || C <= 0 || groups <= 0 || C % groups) return 0;
    return gn_takes_fused(HW, C, groups) ? 1 : 0;
}

extern "C" int sg_groupnorm_nhwc_f16(const sg_groupnorm_desc* d, sg_stream_t stream) {
    SG_REQUIRE(d != nullptr, "sg_groupnorm: null descriptor");
    SG_REQUIRE((d->x || d->split_ws) && d->y && d->gamma && d->beta && d->workspace, "sg_groupnorm: null pointer");
    SG_REQUIRE(d->B > 0 && d->HW > 0 && d->C > 0 && d->groups > 0, "sg_groupnorm: bad shape");
    SG_REQUIRE(d->C % 8 == 0 && d->C % d->groups == 0 && d->C <= 2560,
               "sg_groupnorm: C=%d must be a multiple of 8 and of groups, <= 2560", d->C);
    SG_REQUIRE(d->groups <= GN_MAX_GROUPS, "sg_groupnorm: at most %d groups", GN_MAX_GROUPS);
    SG_REQUIRE(d->ldy % 8 == 0 && d->ldy >= d->C && (d->split_ws || (d->ldx % 8 == 0 && d->ldx >= d->C)), "sg_groupnorm: bad ldx/ldy");
    SG_REQUIRE((d->split_ws || sg_aligned16(d->x)) && sg_aligned16(d->y) && sg_aligned16(d->gamma) && sg_aligned16(d->beta),
               "sg_groupnorm: 16-byte alignment");
    SG_REQUIRE(d->y_pad_w >= 0 && (d->y_pad_w == 0 || d->HW % d->y_pad_w == 0), "sg_groupnorm: y_pad_w must divide HW");
    SG_REQUIRE(!d->xcopy || (sg_aligned16(d->xcopy) && d->ldxc % 8 == 0 && d->ldxc >= d->C), "sg_groupnorm: xcopy alignment / ld");
    SG_REQUIRE(d->workspace_bytes >= sg_groupnorm_workspace_bytes(d->B, d->groups), "sg_groupnorm: workspace too small");
    GnParams p{};
    p.x = d->x; p.ldx = d->ldx; p.x_f32 = d->x_f32 ? 1 : 0;
    p.y = reinterpret_cast<f16*>(d->y); p.ldy = d->ldy; p.pad_w = d->y_pad_w;
    p.xcopy = reinterpret_cast<f16*>(d->xcopy); p.ldxc = d->ldxc;
    p.gamma = reinterpret_cast<const f16*>(d->gamma); p.beta = reinterpret_cast<const f16*>(d->beta);
    p.HW = d->HW; p.C = d->C; p.G = d->groups; p.cpg = d->C / d->groups; p.eps = d->eps; p.silu = d->silu;
    p.ws = reinterpret_cast<float*>(d->workspace);
    if (d->pstats[0]) {
        int covered = 0;
        for (int i = 0; i < 2 && d->pstats[i]; ++i) {
            SG_REQUIRE(d->pstats_rows[i] > 0 && d->HW % d->pstats_rows[i] == 0 && d->pstats_nc[i] > 0 && d->pstats_c0[i] == covered,
                       "sg_groupnorm: pstats[%d]: rows must divide HW and the sources must tile the channels in order", i);
            p.ps[i] = d->pstats[i]; p.ps_rows[i] = d->pstats_rows[i]; p.ps_c0[i] = d->pstats_c0[i]; p.ps_nc[i] = d->pstats_nc[i];
            covered += d->pstats_nc[i];
        }
        SG_REQUIRE(covered == d->C, "sg_groupnorm: pstats cover %d of %d channels", covered, d->C);
    }
    hipStream_t st0 = (hipStream_t)stream;
    // development options (sg_debug_set_option): gn_no_fused, gn_fused_max = <slab elements>, gn_wide
    const SgOptions& opt = sg_options();
    const bool no_fused = opt.gn_no_fused != 0, wide = opt.gn_wide != 0;
    const long fused_max = opt.gn_fused_max >= 0 ? opt.gn_fused_max : GNF_DEFAULT_MAX;
    const long slab = (long)p.HW * p.cpg;
    (void)fused_max; (void)no_fused; (void)slab;
    if (d->split_ws) {
        SG_REQUIRE(gn_takes_fused(p.HW, p.C, p.G), "sg_groupnorm: split_ws needs the one-launch variant (sg_groupnorm_is_fused)");
        SG_REQUIRE(d->split_count >= 2 && d->split_count <= 64 && sg_aligned16(d->split_ws), "sg_groupnorm: split_count in [2, 64], aligned split_ws");
        SG_REQUIRE(!d->pstats[0], "sg_groupnorm: split_ws excludes pstats");
        SG_REQUIRE(!d->split_bias || sg_aligned16(d->split_bias), "sg_groupnorm: split_bias alignment");
        SG_REQUIRE(!d->split_rowbias || (sg_aligned16(d->split_rowbias) && d->split_rowbias_ld % 4 == 0), "sg_groupnorm: split_rowbias alignment");
        SG_REQUIRE(!d->split_res || (sg_aligned16(d->split_res) && d->split_ldr % 8 == 0 && d->split_ldr >= d->C), "sg_groupnorm: split_res alignment / ld");
        SG_REQUIRE(!d->split_out || (sg_aligned16(d->split_out) && d->split_ldo % 8 == 0 && d->split_ldo >= d->C), "sg_groupnorm: split_out alignment / ld");
        p.sp_ws = d->split_ws; p.sp_splits = d->split_count; p.sp_MN = (long)d->B * d->HW * d->C;
        p.sp_bias = reinterpret_cast<const f16*>(d->split_bias);
        p.sp_rowbias = d->split_rowbias; p.sp_rowbias_ld = d->split_rowbias_ld;
        p.sp_res = d->split_res; p.sp_ldr = d->split_ldr; p.sp_res_f32 = d->split_res_f32 ? 1 : 0;
        p.sp_out = d->split_out; p.sp_ldo = d->split_ldo; p.sp_out_f32 = d->split_out_f32 ? 1 : 0;
        p.sp_round = (d->split_out ? !d->split_out_f32 : d->split_round_f16 != 0) ? 1 : 0;
        SG_REQUIRE((long)p.HW * p.cpg <= 1024L * 4 * 4, "sg_groupnorm: split_ws serves slabs of up to 16384 values");
        hipLaunchKernelGGL((gn_fused_kernel<true, 1024, 4>), dim3(p.G, d->B), dim3(1024), 0, st0, p);
        SG_CHECK_LAUNCH("gn_fused<split>");
        return SG_OK;
    }
    if (gn_takes_fused(p.HW, p.C, p.G)) {
        // development option gn_fused_nt = 1024: the 1024-thread instantiation for slabs it can hold (A/B; default 256 threads)
        if (opt.gn_fused_nt == 1024 && (long)p.HW * p.cpg <= 1024L * 4 * 4)
            hipLaunchKernelGGL((gn_fused_kernel<false, 1024, 4>), dim3(p.G, d->B), dim3(1024), 0, st0, p);
        else
            hipLaunchKernelGGL((gn_fused_kernel<false, 256, GNF_MAXI>), dim3(p.G, d->B), dim3(256), 0, st0, p);
        SG_CHECK_LAUNCH("gn_fused");
        return SG_OK;
    }
    if (wide && p.cpg >= 8 && p.C / 8 <= GNW_MAX_VPR) {
        const int rpp = GNW_NT / (p.C / 8);
        // One 1024-thread workgroup fits a CU (99 VGPRs), so the launch is ONE round of at most 256 workgroups: 256 / B chunks per
        // sample (B = 3: 85 chunks of 49 rows = two full passes of the 25 thread rows at 64^2 x 320, where the round-1..3 cap of 64 chunks
        // left 64 CUs idle and a third, half-empty pass; 128 chunks at B = 3 — a second round — measured 1.6x slower per launch)
        int want = 256 / d->B;
        if (opt.gn_chunks > 0 && want > opt.gn_chunks) want = opt.gn_chunks;     // development option (A/B): 64 = the round-1..3 cap
        if (want < 1) want = 1;
        if (want > GNW_MAX_CHUNKS) want = GNW_MAX_CHUNKS;
        p.rows_per_chunk = sg_cdiv(p.HW, want);
        if (p.rows_per_chunk < rpp) p.rows_per_chunk = rpp;   // at least one full pass of the thread rows
        p.nchunks = sg_cdiv(p.HW, p.rows_per_chunk);
        const dim3 grid(p.nchunks, d->B), block(GNW_NT);
        if (!p.ps[0]) {            // no statistics from the producers' epilogues: own pass over x
            hipLaunchKernelGGL(gn_stats_wide_kernel, grid, block, 0, st0, p);
            SG_CHECK_LAUNCH("gn_stats_wide");
        }
        hipLaunchKernelGGL(gn_apply_wide_kernel, grid, block, 0, st0, p);
        SG_CHECK_LAUNCH("gn_apply_wide");
        return SG_OK;
    }
    p.ps[0] = p.ps[1] = nullptr;       // the narrow pair keeps its own statistics pass
    int apply_blocks = 1;
    gn_geometry(d->B, d->HW, d->C, d->x_f32 != 0, &p.nchunks, &p.rows_per_chunk, &apply_blocks, &p.apply_rows);
    dim3 grid(p.nchunks, d->B), block(256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_stats_kernel, grid, block, 0, st, p);
    SG_CHECK_LAUNCH("gn_stats");
    hipLaunchKernelGGL(gn_apply_kernel, dim3(apply_blocks, d->B), block, 0, st, p);
    SG_CHECK_LAUNCH("gn_apply");
    return SG_OK;
}

extern "C" int sg_layernorm_f16(const void* x, int64_t ldx, int32_t x_f32, int32_t M, int32_t C, float eps, const sg_half* gamma1,
                                const sg_half* beta1, sg_half* y1, int64_t ldy1, const sg_half* gamma2,
                                const sg_half* beta2, sg_half* y2, int64_t ldy2, sg_stream_t stream) {
    SG_REQUIRE(x && gamma1 && beta1 && y1, "sg_layernorm: null pointer");
    SG_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "sg_layernorm: C=%d must be a multiple of 8, <= 2048", C);
    SG_REQUIRE(ldx % 8 == 0 && ldy1 % 8 == 0 && ldx >= C && ldy1 >= C, "sg_layernorm: bad ldx/ldy1");
    SG_REQUIRE(sg_aligned16(x) && sg_aligned16(y1) && sg_aligned16(gamma1) && sg_aligned16(beta1), "sg_layernorm: 16-byte alignment");
    if (y2) {
        SG_REQUIRE(gamma2 && beta2 && sg_aligned16(y2) && sg_aligned16(gamma2) && sg_aligned16(beta2) && ldy2 % 8 == 0 &&
                       ldy2 >= C, "sg_layernorm: second output arguments");
    }
    LnParams p{};
    p.x = x; p.ldx = ldx; p.x_f32 = x_f32 ? 1 : 0; p.M = M; p.C = C; p.eps = eps;
    p.g1 = reinterpret_cast<const f16*>(gamma1); p.b1 = reinterpret_cast<const f16*>(beta1);
    p.y1 = reinterpret_cast<f16*>(y1); p.ldy1 = ldy1;
    p.g2 = reinterpret_cast<const f16*>(gamma2); p.b2 = reinterpret_cast<const f16*>(beta2);
    p.y2 = reinterpret_cast<f16*>(y2); p.ldy2 = ldy2;
    dim3 grid(sg_cdiv(M, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int nv = sg_cdiv(C / 8, 64);
    if (nv == 1) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, p);
    else if (nv == 2) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, st, p);
    else if (nv == 3) hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, st, p);
    SG_CHECK_LAUNCH("sg_layernorm_f16");
    return SG_OK;
}

extern "C" size_t sg_groupnorm_bwd_workspace_bytes(int32_t B, int32_t groups) {
    return 2 * sg_groupnorm_workspace_bytes(B, groups);     // statistics partials of x | partials of (g, g * xhat)
}

extern "C" int sg_groupnorm_bwd_nhwc_f16(const sg_groupnorm_bwd_desc* d, sg_stream_t stream) {
    SG_REQUIRE(d != nullptr, "sg_groupnorm_bwd: null descriptor");
    SG_REQUIRE(d->x && d->dy && d->gamma && d->beta && d->workspace && d->out, "sg_groupnorm_bwd: null pointer");
    SG_REQUIRE(d->B > 0 && d->HW > 0 && d->C > 0 && d->groups > 0 && d->groups <= GN_MAX_GROUPS, "sg_groupnorm_bwd: bad shape");
    SG_REQUIRE(d->C % 8 == 0 && d->C % d->groups == 0, "sg_groupnorm_bwd: C=%d must be a multiple of 8 and of groups", d->C);
    const int cpg = d->C / d->groups;
    if (cpg < 8 || d->C / 8 > GNW_MAX_VPR)
        return sg_set_error(SG_EUNSUP, "sg_groupnorm_bwd: needs >= 8 channels per group and C <= %d (got C=%d, groups=%d)",
                            GNW_MAX_VPR * 8, d->C, d->groups);
    SG_REQUIRE(d->ldx % 8 == 0 && d->lddy % 8 == 0 && d->ldo % 8 == 0 && d->ldx >= d->C && d->lddy >= d->C && d->ldo >= d->C,
               "sg_groupnorm_bwd: bad ld");
    SG_REQUIRE(sg_aligned16(d->x) && sg_aligned16(d->dy) && sg_aligned16(d->out) && sg_aligned16(d->gamma) && sg_aligned16(d->beta),
               "sg_groupnorm_bwd: 16-byte alignment");
    SG_REQUIRE(!d->res || (d->out_f32 && sg_aligned16(d->res) && d->ldr % 8 == 0 && d->ldr >= d->C),
               "sg_groupnorm_bwd: a residual gradient needs the fp32 output, 16-byte alignment and ldr >= C");
    SG_REQUIRE(d->out_pad_w >= 0 && (d->out_pad_w == 0 || (!d->out_f32 && d->HW % d->out_pad_w == 0)),
               "sg_groupnorm_bwd: out_pad_w needs the fp16 output and must divide HW");
    SG_REQUIRE(d->workspace_bytes >= sg_groupnorm_bwd_workspace_bytes(d->B, d->groups), "sg_groupnorm_bwd: workspace too small");
    GnBwdParams q{};
    GnParams& p = q.f;
    p.x = d->x; p.ldx = d->ldx; p.x_f32 = d->x_f32 ? 1 : 0;
    p.gamma = reinterpret_cast<const f16*>(d->gamma); p.beta = reinterpret_cast<const f16*>(d->beta);
    p.HW = d->HW; p.C = d->C; p.G = d->groups; p.cpg = cpg; p.eps = d->eps; p.silu = d->silu;
    p.ws = reinterpret_cast<float*>(d->workspace);
    q.ws2 = p.ws + (size_t)d->B * GN_MAX_CHUNKS * d->groups * 2;
    q.dy = d->dy; q.lddy = d->lddy; q.dy_f32 = d->dy_f32 ? 1 : 0;
    q.res = d->res; q.ldr = d->ldr;
    if (d->out_f32) { q.out32 = reinterpret_cast<float*>(d->out); q.ldo32 = d->ldo; }
    else { q.out16 = reinterpret_cast<f16*>(d->out); q.ldo16 = d->ldo; q.pad_w = d->out_pad_w; }
    const int rpp = GNW_NT / (p.C / 8);
    int want = sg_cdiv(320, d->B);
    if (want > GN_MAX_CHUNKS) want = GN_MAX_CHUNKS;
    p.rows_per_chunk = sg_cdiv(p.HW, want);
    if (p.rows_per_chunk < rpp) p.rows_per_chunk = rpp;
    p.nchunks = sg_cdiv(p.HW, p.rows_per_chunk);
    const dim3 grid(p.nchunks, d->B), block(GNW_NT);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_stats_wide_kernel, grid, block, 0, st, p);
    SG_CHECK_LAUNCH("gn_bwd: stats");
    hipLaunchKernelGGL(gn_bwd_sums_kernel, grid, block, 0, st, q);
    SG_CHECK_LAUNCH("gn_bwd: sums");
    hipLaunchKernelGGL(gn_bwd_apply_kernel, grid, block, 0, st, q);
    SG_CHECK_LAUNCH("gn_bwd: apply");
    return SG_OK;
}
