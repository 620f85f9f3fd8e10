// Small kernels of the hot path: library bookkeeping, time embedding, the two thin 3x3 convolutions (4->C and
// C->4 channels, bandwidth/latency-bound so no MFMA), the sampling-loop elementwise steps and the strided row copy.
#include "common.h"
#include <string.h>

// ------------------------------------------------------------------------------------------------ bookkeeping
static thread_local char g_err[512] = "";

SgOptions& sg_options() {
    static SgOptions o;
    return o;
}

extern "C" int sg_debug_set_option(const char* name, int64_t value) {
    if (!name) return sg_set_error(SG_EINVAL, "sg_debug_set_option: null name");
    SgOptions& o = sg_options();
    struct Entry { const char* n; int* p; };
    const Entry table[] = {{"tile_m", &o.tile_m}, {"tile_n", &o.tile_n}, {"no_pipe", &o.no_pipe}, {"no_split", &o.no_split},
                           {"no_nmajor", &o.no_nmajor}, {"attn_sub2", &o.attn_sub2}, {"attn_prio", &o.attn_prio},
                           {"attn_d80", &o.attn_d80}, {"attn_d160", &o.attn_d160}, {"gn_no_fused", &o.gn_no_fused},
                           {"gn_wide", &o.gn_wide}, {"attn_lean", &o.attn_lean}, {"attn_d40_general", &o.attn_d40_general},
                           {"gn_fused_nt", &o.gn_fused_nt}, {"pipe_stages", &o.pipe_stages}, {"ff_variant", &o.ff_variant},
                           {"gn_chunks", &o.gn_chunks}, {"lat_tiles", &o.lat_tiles}, {"lat_min_kt", &o.lat_min_kt},
                           {"lat_max_kt", &o.lat_max_kt}, {"lat_stages", &o.lat_stages}, {"lat_wide", &o.lat_wide}, {"lat_mask", &o.lat_mask}, {"lat_wide_m", &o.lat_wide_m}, {"fat_m", &o.fat_m}, {"big_m", &o.big_m}, {"big_bm", &o.big_bm}, {"big_bn", &o.big_bn}};
    for (const Entry& e : table)
        if (strcmp(e.n, name) == 0) {
            *e.p = (int)value;
            return SG_OK;
        }
    if (strcmp(name, "gn_fused_max") == 0) {
        o.gn_fused_max = (long)value;
        return SG_OK;
    }
    if (strcmp(name, "reset") == 0) {
        o = SgOptions{};
        return SG_OK;
    }
    return sg_set_error(SG_EINVAL, "sg_debug_set_option: unknown option '%s'", name);
}

int sg_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int sg_version(void) { return 1; }
extern "C" const char* sg_last_error(void) { return g_err; }

extern "C" int sg_device_arch(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return sg_set_error(SG_ELAUNCH, "sg_device_arch: no HIP device");
    const char* name = prop.gcnArchName;   // e.g. "gfx950:sramecc+:xnack-"
    if (strncmp(name, "gfx", 3) != 0) return sg_set_error(SG_EARCH, "sg_device_arch: unexpected arch '%s'", name);
    return (int)strtol(name + 3, nullptr, 10);
}

extern "C" int sg_device_cus(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return sg_set_error(SG_ELAUNCH, "sg_device_cus: no HIP device");
    return prop.multiProcessorCount;
}

namespace {

// ------------------------------------------------------------------------------------------------ time embedding
__global__ void timestep_embed_kernel(const float* t, const float* freqs, float* out, int B, int dim, int flip) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * half) return;
    const int b = idx / half, j = idx - b * half;
    const float e = t[b] * freqs[j];
    const float s = sinf(e), c = cosf(e);
    float* o = out + (long)b * dim;
    if (flip) { o[j] = c; o[half + j] = s; } else { o[j] = s; o[half + j] = c; }
}

// y[b][n] = act_out(sum_k act_in(x[b][k]) W[n][k] + bias[n]); one wave per output column n, all B rows at once
// (the weight row is streamed exactly once: this is a weight-bandwidth-bound GEMV bundle).
template <int MAXB>
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* x, long ldx, const f16* W, long ldw, const f16* bias,
                                                          float* y, long ldy, int B, int N, int K, int act_in, int act_out) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[MAXB];
#pragma unroll
    for (int b = 0; b < MAXB; ++b) acc[b] = 0.f;
    const f16* wr = W + (long)n * ldw;
    for (int k0 = lane * 8; k0 < K; k0 += 64 * 8) {
        H8 w; w.u = ldg16(wr + k0);
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if (b < B) {
                const float4 x0 = *reinterpret_cast<const float4*>(x + b * ldx + k0);
                const float4 x1 = *reinterpret_cast<const float4*>(x + b * ldx + k0 + 4);
                float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = act_in ? silu_f(xv[j]) : xv[j];
                    acc[b] += a * (float)w.h[j];
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        if (b < B) {
            float v = wave_sum(acc[b]);
            if (lane == 0) {
                if (bias) v += (float)bias[n];
                if (act_out) v = silu_f(v);
                y[b * ldy + n] = v;
            }
        }
    }
}

// out[b][:] = table[j][:] for the j with table_keys[j] == keys[b] (exact comparison; NaN rows when no key matches, so a stale table
// cannot go unnoticed).  One workgroup per (row b, 1024-float column block); T is a few dozen: every thread scans the keys itself.
__global__ __launch_bounds__(256) void lookup_rows_kernel(const float* keys, const float* table_keys, int T, const float* table,
                                                          long ldt, float* out, long ldo, int N) {
    const int b = blockIdx.y;
    const float key = keys[b];
    int j = -1;
    for (int i = 0; i < T; ++i)
        if (j < 0 && table_keys[i] == key) j = i;
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (n >= N) return;
    float4 v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
    if (j >= 0) v = *reinterpret_cast<const float4*>(table + (long)j * ldt + n);
    *reinterpret_cast<float4*>(out + (long)b * ldo + n) = v;
}

// ------------------------------------------------------------------------------------------------ conv_in / conv_out
// conv_in: thread = (pixel, 8 output channels); x fp32 NCHW, w fp16 [9*Cin][Cout].
__global__ __launch_bounds__(256) void conv_in_kernel(const float* x, const f16* w, const f16* bias, void* y, long ldy,
                                                      int y_f32, int B, int H, int Wd, int Cin, int Cout) {
    const int cg = Cout / 8;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)B * H * Wd * cg;
    if (idx >= total) return;
    const int g = (int)(idx % cg);
    const long pix = idx / cg;
    const int ox = (int)(pix % Wd), oy = (int)((pix / Wd) % H), b = (int)(pix / ((long)Wd * H));
    float acc[8];
    {
        H8 bb; bb.u = ldg16(bias + g * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = (float)bb.h[j];
    }
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + ky - 1;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox + kx - 1;
            if ((unsigned)ix >= (unsigned)Wd) continue;
            for (int ci = 0; ci < Cin; ++ci) {
                const float xv = x[(((long)b * Cin + ci) * H + iy) * Wd + ix];
                H8 wv; wv.u = ldg16(w + ((long)((ky * 3 + kx) * Cin + ci)) * Cout + g * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += xv * (float)wv.h[j];
            }
        }
    }
    if (y_f32) {
        float* o = reinterpret_cast<float*>(y) + pix * ldy + g * 8;
        *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    } else {
        store8h(reinterpret_cast<f16*>(y) + pix * ldy + g * 8, acc);
    }
}

// conv_out: one wave per output pixel; lanes split the 9*Cin/8 input chunks; Cout <= 4 accumulators.
__global__ __launch_bounds__(256) void conv_out_kernel(const f16* x, long ldx, const f16* w, const f16* bias, float* y,
                                                       int B, int H, int Wd, int Cin, int Cout) {
    const int lane = threadIdx.x & 63;
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (long)B * H * Wd) return;
    const int ox = (int)(pix % Wd), oy = (int)((pix / Wd) % H), b = (int)(pix / ((long)Wd * H));
    const int cpt = Cin / 8, nchunk = 9 * cpt;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ci = lane; ci < nchunk; ci += 64) {
        const int tap = ci / cpt, cc = ci - tap * cpt;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)Wd) continue;
        H8 xv; xv.u = ldg16(x + (((long)b * H + iy) * Wd + ix) * ldx + cc * 8);
        for (int co = 0; co < Cout; ++co) {
            H8 wv; wv.u = ldg16(w + ((long)co * 9 + tap) * Cin + cc * 8);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)xv.h[j] * (float)wv.h[j];
            acc[co] += s;
        }
    }
    for (int co = 0; co < Cout; ++co) {
        const float v = wave_sum(acc[co]);
        if (lane == 0) y[(((long)b * Cout + co) * H + oy) * Wd + ox] = v + (float)bias[co];
    }
}

// ------------------------------------------------------------------------------------------------ sampling-loop steps
// out[u] = coef[2u] * src[u] + coef[2u+1] * noise[u % N]   (blockIdx.y = u)
__global__ void add_noise_kernel(const float* src, const float* noise, const float* coef, float* out, int N, long n) {
    const int u = blockIdx.y;
    const float a = coef[2 * u], s = coef[2 * u + 1];
    const float* x = src + (long)u * n;
    const float* z = noise + (long)(u % N) * n;
    float* o = out + (long)u * n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        o[i] = a * x[i] + s * z[i];
}

__global__ void cfg_ddim_kernel(const float* eps3, float* lat, float* lat3, const float* coef, int N, long n) {
    const long total = (long)N * n;
    const float s_img = coef[0], s_txt = coef[1], sa = coef[2], sb = coef[3], sap = coef[4], sbp = coef[5];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float eu = eps3[i], ei = eps3[total + i], ea = eps3[2 * total + i];
        const float eps = eu + s_img * (ei - eu) + s_txt * (ea - ei);
        const float x = lat[i];
        const float x0 = (x - sb * eps) / sa;
        const float xp = sap * x0 + sbp * eps;
        lat[i] = xp;
        if (lat3) { lat3[i] = xp; lat3[total + i] = xp; lat3[2 * total + i] = xp; }
    }
}

// PNDM / PLMS update (diffusers PNDMScheduler.step_plms + _get_prev_sample, skip_prk_steps): coef = [s_img, s_txt, A, Bc,
// w0..w3, slot_cur, slot1..slot3, push, use_kept, keep] (storygen_amd/scheduler.py::PNDMSchedule.step_row).
__global__ void cfg_plms_kernel(const float* eps3, float* lat, float* lat3, float* hist, float* kept, const float* coef, int N,
                                long n) {
    const long total = (long)N * n;
    const float s_img = coef[0], s_txt = coef[1], A = coef[2], Bc = coef[3];
    const float w0 = coef[4], w1 = coef[5], w2 = coef[6], w3 = coef[7];
    const int cur = (int)coef[8], s1 = (int)coef[9], s2 = (int)coef[10], s3 = (int)coef[11];
    const bool push = coef[12] != 0.f, use_kept = coef[13] != 0.f, keep = coef[14] != 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float eu = eps3[i], ei = eps3[total + i], ea = eps3[2 * total + i];
        const float e = eu + s_img * (ei - eu) + s_txt * (ea - ei);
        // history terms are read before the (optional) store: slot_cur never aliases a slot with a non-zero weight
        float ep = w0 * e;
        if (w1 != 0.f) ep += w1 * hist[s1 * total + i];
        if (w2 != 0.f) ep += w2 * hist[s2 * total + i];
        if (w3 != 0.f) ep += w3 * hist[s3 * total + i];
        if (push) hist[cur * total + i] = e;
        const float x = lat[i];
        const float xs = use_kept ? kept[i] : x;
        if (keep) kept[i] = x;
        const float xp = A * xs - Bc * ep;
        lat[i] = xp;
        if (lat3) { lat3[i] = xp; lat3[total + i] = xp; lat3[2 * total + i] = xp; }
    }
}

// mode 0: fp16 -> fp16, 1: fp32 -> fp32, 2: fp32 -> fp16 (cast); 8 elements per thread per iteration
__global__ __launch_bounds__(256) void copy_rows_kernel(void* dst, long ldd, long bsd, const void* src, long lds, long bss,
                                                        int batches, int rows, int cols, int mode) {
    const int vpr = cols / 8;
    const long total = (long)batches * rows * vpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpr);
        const long rr = i / vpr;
        const int r = (int)(rr % rows), b = (int)(rr / rows);
        const long so = b * bss + (long)r * lds + cv * 8, doff = b * bsd + (long)r * ldd + cv * 8;
        if (mode == 0) {
            stg16(reinterpret_cast<f16*>(dst) + doff, ldg16(reinterpret_cast<const f16*>(src) + so));
        } else {
            float v[8];
            load8f(src, so, true, v);
            if (mode == 1) {
                float* o = reinterpret_cast<float*>(dst) + doff;
                *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                store8h(reinterpret_cast<f16*>(dst) + doff, v);
            }
        }
    }
}

// x [B,H,W,C] (fp16 or fp32) -> interior of the zero-bordered fp16 [B,H+2,W+2,C]
__global__ __launch_bounds__(256) void pad_cast_kernel(const void* x, long ldx, int x_f32, f16* y, long ldy, int B, int H, int Wd,
                                                       int C) {
    const int vpr = C / 8;
    const long total = (long)B * H * Wd * vpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % vpr);
        const long pix = i / vpr;
        const int xx = (int)(pix % Wd), yy = (int)((pix / Wd) % H), b = (int)(pix / ((long)Wd * H));
        float v[8];
        load8f(x, pix * ldx + cv * 8, x_f32, v);
        store8h(y + (((long)b * (H + 2) + yy + 1) * (Wd + 2) + xx + 1) * ldy + cv * 8, v);
    }
}

}  // namespace

extern "C" int sg_timestep_embed_f32(const float* t, const float* freqs, float* out, int32_t B, int32_t dim,
                                     int32_t flip_sin_to_cos, sg_stream_t stream) {
    SG_REQUIRE(t && freqs && out, "sg_timestep_embed: null pointer");
    SG_REQUIRE(B > 0 && dim > 0 && dim % 2 == 0, "sg_timestep_embed: bad shape");
    const int total = B * (dim / 2);
    hipLaunchKernelGGL(timestep_embed_kernel, dim3(sg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, t, freqs, out, B,
                       dim, flip_sin_to_cos);
    SG_CHECK_LAUNCH("sg_timestep_embed_f32");
    return SG_OK;
}

extern "C" int sg_linear_rows_f32(const float* x, int64_t ldx, const sg_half* W, int64_t ldw, const sg_half* bias, float* y,
                                  int64_t ldy, int32_t B, int32_t N, int32_t K, int32_t act_in, int32_t act_out,
                                  sg_stream_t stream) {
    SG_REQUIRE(x && W && y, "sg_linear_rows: null pointer");
    SG_REQUIRE(B > 0 && B <= 16 && N > 0 && K > 0 && K % 8 == 0, "sg_linear_rows: bad shape B=%d N=%d K=%d", B, N, K);
    SG_REQUIRE(ldx % 4 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K && ldy >= N, "sg_linear_rows: bad strides");
    SG_REQUIRE(sg_aligned16(x) && sg_aligned16(W), "sg_linear_rows: 16-byte alignment");
    dim3 grid(sg_cdiv(N, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const f16* w = reinterpret_cast<const f16*>(W);
    const f16* bs = reinterpret_cast<const f16*>(bias);
    if (B <= 4) hipLaunchKernelGGL(linear_rows_kernel<4>, grid, block, 0, st, x, (long)ldx, w, (long)ldw, bs, y, (long)ldy, B, N, K, act_in, act_out);
    else if (B <= 8) hipLaunchKernelGGL(linear_rows_kernel<8>, grid, block, 0, st, x, (long)ldx, w, (long)ldw, bs, y, (long)ldy, B, N, K, act_in, act_out);
    else hipLaunchKernelGGL(linear_rows_kernel<16>, grid, block, 0, st, x, (long)ldx, w, (long)ldw, bs, y, (long)ldy, B, N, K, act_in, act_out);
    SG_CHECK_LAUNCH("sg_linear_rows_f32");
    return SG_OK;
}

extern "C" int sg_lookup_rows_f32(const float* keys, int32_t B, const float* table_keys, int32_t T, const float* table, int64_t ldt,
                                  float* out, int64_t ldo, int32_t N, sg_stream_t stream) {
    SG_REQUIRE(keys && table_keys && table && out && B > 0 && T > 0 && N > 0, "sg_lookup_rows: bad arguments");
    SG_REQUIRE(B <= 65535 && N % 4 == 0 && ldt % 4 == 0 && ldo % 4 == 0 && ldt >= N && ldo >= N, "sg_lookup_rows: N and the strides must be multiples of 4");
    SG_REQUIRE(sg_aligned16(table) && sg_aligned16(out), "sg_lookup_rows: 16-byte alignment");
    hipLaunchKernelGGL(lookup_rows_kernel, dim3(sg_cdiv(N, 1024), B), dim3(256), 0, (hipStream_t)stream, keys, table_keys, T, table,
                       (long)ldt, out, (long)ldo, N);
    SG_CHECK_LAUNCH("sg_lookup_rows_f32");
    return SG_OK;
}

extern "C" int sg_conv_in_f16(const float* x_nchw, const sg_half* w_kn, const sg_half* bias, void* y, int64_t ldy,
                              int32_t y_f32, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                              sg_stream_t stream) {
    SG_REQUIRE(x_nchw && w_kn && bias && y, "sg_conv_in: null pointer");
    SG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 8 && Cout > 0 && Cout % 8 == 0, "sg_conv_in: bad shape");
    SG_REQUIRE(ldy % 8 == 0 && ldy >= Cout && sg_aligned16(w_kn) && sg_aligned16(bias) && sg_aligned16(y), "sg_conv_in: alignment");
    const long total = (long)B * H * W * (Cout / 8);
    hipLaunchKernelGGL(conv_in_kernel, dim3(sg_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                       reinterpret_cast<const f16*>(w_kn), reinterpret_cast<const f16*>(bias), y, (long)ldy, y_f32, B, H, W,
                       Cin, Cout);
    SG_CHECK_LAUNCH("sg_conv_in_f16");
    return SG_OK;
}

extern "C" int sg_conv_out_f16(const sg_half* x, int64_t ldx, const sg_half* w, const sg_half* bias, float* y_nchw, int32_t B,
                               int32_t H, int32_t W, int32_t Cin, int32_t Cout, sg_stream_t stream) {
    SG_REQUIRE(x && w && bias && y_nchw, "sg_conv_out: null pointer");
    SG_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout <= 4, "sg_conv_out: bad shape");
    SG_REQUIRE(ldx % 8 == 0 && ldx >= Cin && sg_aligned16(x) && sg_aligned16(w), "sg_conv_out: alignment");
    const long pixels = (long)B * H * W;
    hipLaunchKernelGGL(conv_out_kernel, dim3(sg_cdiv(pixels, 4)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f16*>(x), (long)ldx, reinterpret_cast<const f16*>(w),
                       reinterpret_cast<const f16*>(bias), y_nchw, B, H, W, Cin, Cout);
    SG_CHECK_LAUNCH("sg_conv_out_f16");
    return SG_OK;
}

extern "C" int sg_add_noise_f32(const float* src, const float* noise, const float* coef, float* out, int32_t U, int32_t N,
                                int64_t n, sg_stream_t stream) {
    SG_REQUIRE(src && noise && coef && out && U > 0 && N > 0 && n > 0, "sg_add_noise: bad arguments");
    SG_REQUIRE(U <= 65535, "sg_add_noise: at most 65535 samples");
    hipLaunchKernelGGL(add_noise_kernel, dim3((int)min((long)256, (n + 255) / 256), U), dim3(256), 0, (hipStream_t)stream, src,
                       noise, coef, out, N, (long)n);
    SG_CHECK_LAUNCH("sg_add_noise_f32");
    return SG_OK;
}

extern "C" int sg_cfg_ddim_step_f32(const float* eps3, float* latents, float* latents3, const float* coef, int32_t N, int64_t n,
                                    sg_stream_t stream) {
    SG_REQUIRE(eps3 && latents && coef && N > 0 && n > 0, "sg_cfg_ddim_step: bad arguments");
    const long total = (long)N * n;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3((int)min((long)1024, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, eps3,
                       latents, latents3, coef, N, (long)n);
    SG_CHECK_LAUNCH("sg_cfg_ddim_step_f32");
    return SG_OK;
}

extern "C" int sg_cfg_plms_step_f32(const float* eps3, float* latents, float* latents3, float* history, float* kept,
                                    const float* coef, int32_t N, int64_t n, sg_stream_t stream) {
    SG_REQUIRE(eps3 && latents && history && kept && coef && N > 0 && n > 0, "sg_cfg_plms_step: bad arguments");
    const long total = (long)N * n;
    hipLaunchKernelGGL(cfg_plms_kernel, dim3((int)min((long)1024, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, eps3,
                       latents, latents3, history, kept, coef, N, (long)n);
    SG_CHECK_LAUNCH("sg_cfg_plms_step_f32");
    return SG_OK;
}

extern "C" int sg_copy_rows(void* dst, int64_t ldd, int64_t bsd, const void* src, int64_t lds, int64_t bss, int32_t batches,
                            int32_t rows, int32_t cols, int32_t mode, sg_stream_t stream) {
    SG_REQUIRE(dst && src && batches > 0 && rows > 0 && cols > 0, "sg_copy_rows: bad arguments");
    SG_REQUIRE(mode >= 0 && mode <= 2, "sg_copy_rows: mode must be 0 (f16), 1 (f32) or 2 (f32->f16)");
    SG_REQUIRE(cols % 8 == 0 && ldd % 8 == 0 && lds % 8 == 0 && bsd % 8 == 0 && bss % 8 == 0, "sg_copy_rows: multiples of 8");
    SG_REQUIRE(sg_aligned16(dst) && sg_aligned16(src), "sg_copy_rows: 16-byte alignment");
    const long total = (long)batches * rows * (cols / 8);
    hipLaunchKernelGGL(copy_rows_kernel, dim3((int)min((long)2048, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst,
                       (long)ldd, (long)bsd, src, (long)lds, (long)bss, batches, rows, cols, mode);
    SG_CHECK_LAUNCH("sg_copy_rows");
    return SG_OK;
}

extern "C" int sg_pad_cast_f16(const void* x, int64_t ldx, int32_t x_f32, sg_half* y, int64_t ldy, int32_t B, int32_t H,
                               int32_t W, int32_t C, sg_stream_t stream) {
    SG_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0, "sg_pad_cast: bad arguments");
    SG_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C, "sg_pad_cast: multiples of 8");
    SG_REQUIRE(sg_aligned16(x) && sg_aligned16(y), "sg_pad_cast: 16-byte alignment");
    const long total = (long)B * H * W * (C / 8);
    hipLaunchKernelGGL(pad_cast_kernel, dim3((int)min((long)2048, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (long)ldx, x_f32, reinterpret_cast<f16*>(y), (long)ldy, B, H, W, C);
    SG_CHECK_LAUNCH("sg_pad_cast_f16");
    return SG_OK;
}
