// Backward-pass kernels of the stage-2 training step (BASELINE config 4, train_StorySalon_stage2.py:322-327): the
// bandwidth-bound ones.  Formulas and layer order: oracle/storygen_backward.py (checked against torch.autograd and the
// reference's gradients on the CPU).  The contractions of the backward pass reuse the forward kernels: a linear layer's
// dgrad is sg_gemm_f16 with the transposed weight, its weight gradient is sg_gemm_f16 on transposed activations
// (sg_transpose_f16 below), a convolution's dgrad is sg_conv3x3_nhwc_f16 with the 180-degree-rotated, channel-swapped
// weight (stride 2: on the zero-stuffed gradient, sg_zero_stuff_f16; nearest-2x upsampling: followed by sg_sum2x2).
//
// STATUS: validated on MI355X in round 2 (tests/test_backward_gpu.py, every kernel green on its first hardware run).
#include "common.h"

namespace {

__device__ __forceinline__ float gelu_grad_f(float x) {   // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = __expf(-0.5f * x * x) * 0.39894228040143267794f;
    return cdf + x * pdf;
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
struct LnBwdParams {
    const void* x; long ldx; int x_f32;
    const void* dy1; long lddy1; const f16* g1;
    const void* dy2; long lddy2; const f16* g2;     // optional second (dy, gamma) pair sharing x (norm2 / norm4)
    int dy_f32;
    const float* res; long ldr; float res_scale;    // optional: out = res_scale * res + dx
    float* out; long ldo;
    int M, C; float eps;
};

// One wave per row, the row stays in registers (C <= 64 * 8 * NV).  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),
// g = dy1 * gamma1 (+ dy2 * gamma2): the formula is linear in g, so two LayerNorms of the same input share one pass.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const LnBwdParams p) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int vpr = p.C / 8;
    float x[NV][8], g[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cv = lane + 64 * i;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[i][j] = g[i][j] = 0.f;
        if (cv < vpr) {
            load8f(p.x, (long)row * p.ldx + cv * 8, p.x_f32, x[i]);
            float d[8];
            H8 gm;
            load8f(p.dy1, (long)row * p.lddy1 + cv * 8, p.dy_f32, d);
            gm.u = ldg16(p.g1 + cv * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[i][j] = d[j] * (float)gm.h[j];
            if (p.dy2) {
                load8f(p.dy2, (long)row * p.lddy2 + cv * 8, p.dy_f32, d);
                gm.u = ldg16(p.g2 + cv * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) g[i][j] += d[j] * (float)gm.h[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += x[i][j];
        }
    }
    const float inv_c = 1.0f / (float)p.C;
    const float mean = wave_sum(sum) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < vpr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = x[i][j] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(sq) * inv_c + p.eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < vpr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[i][j] = (x[i][j] - mean) * rstd;      // xhat
                s1 += g[i][j];
                s2 += g[i][j] * x[i][j];
            }
        }
    const float m1 = wave_sum(s1) * inv_c, m2 = wave_sum(s2) * inv_c;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int cv = lane + 64 * i;
        if (cv < vpr) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rstd * (g[i][j] - m1 - x[i][j] * m2);
            if (p.res) {
                const float* r = p.res + (long)row * p.ldr + cv * 8;
                const float4 a = *reinterpret_cast<const float4*>(r), b = *reinterpret_cast<const float4*>(r + 4);
                o[0] += p.res_scale * a.x; o[1] += p.res_scale * a.y; o[2] += p.res_scale * a.z; o[3] += p.res_scale * a.w;
                o[4] += p.res_scale * b.x; o[5] += p.res_scale * b.y; o[6] += p.res_scale * b.z; o[7] += p.res_scale * b.w;
            }
            float* q = p.out + (long)row * p.ldo + cv * 8;
            *reinterpret_cast<float4*>(q) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(q + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ GEGLU backward
// proj / dproj: [M, N8] in the 32/32-interleaved column layout the forward GEMM's weight rows have (64-column blocks of
// 32 value columns followed by their 32 gate columns, see epi_geglu8 in gemm_conv.hip); du: [M, N8 / 2].
// dval = du * gelu(gate), dgate = du * val * gelu'(gate).
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const f16* proj, long ldp, const f16* du, long ldu, f16* dproj, long lddp,
                                                        int M, int N8) {
    const int och = N8 / 16;    // 8-column chunks of du per row
    const long total = (long)M * och;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int m = (int)(idx / och), j = (int)(idx - (long)m * och);
        const int vcol = (j >> 2) * 64 + (j & 3) * 8;         // value columns; gates at vcol + 32
        H8 v, g, d, ov, og;
        v.u = ldg16(proj + (long)m * ldp + vcol);
        g.u = ldg16(proj + (long)m * ldp + vcol + 32);
        d.u = ldg16(du + (long)m * ldu + j * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gate = (float)g.h[e], val = (float)v.h[e], dd = (float)d.h[e];
            ov.h[e] = (f16)(dd * gelu_erf_f(gate));
            og.h[e] = (f16)(dd * val * gelu_grad_f(gate));
        }
        stg16(dproj + (long)m * lddp + vcol, ov.u);
        stg16(dproj + (long)m * lddp + vcol + 32, og.u);
    }
}

// ------------------------------------------------------------------------------------------------ transpose
// dst[c][m] = src[m][c] (fp16 out; fp16 or fp32 in): 64 x 64 tiles through LDS (padded rows), 16-byte global accesses on
// both sides.  Used for the weight gradients: dW[n, k] = sum_m dy[m, n] x[m, k] is sg_gemm_f16(A = dy^T, W = x^T).
// blockIdx.z = batch (sg_transpose_batched_f16: the B images of an attention operand in ONE launch; bss / bsd = batch strides in elements)
__global__ __launch_bounds__(256) void transpose_kernel(const void* src0, long lds_, long bss, int src_f32, f16* dst0, long ldd, long bsd,
                                                        int M, int C) {
    __shared__ f16 tile[64][72];
    const int t = threadIdx.x;
    const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const void* src = src_f32 ? static_cast<const void*>(static_cast<const float*>(src0) + (long)blockIdx.z * bss)
                              : static_cast<const void*>(static_cast<const f16*>(src0) + (long)blockIdx.z * bss);
    f16* dst = dst0 + (long)blockIdx.z * bsd;
    {   // load: 64 rows x 8 chunks of 8 columns; thread -> (row = t / 8 + 32 * i, chunk = t % 8)
        const int ch = t & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (t >> 3) + 32 * i;
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (m0 + r < M && c0 + ch * 8 < C) load8f(src, (long)(m0 + r) * lds_ + c0 + ch * 8, src_f32, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[r][ch * 8 + e] = (f16)v[e];
        }
    }
    __syncthreads();
    {   // store: 64 output rows (columns of src) x 8 chunks of 8 consecutive m
        const int ch = t & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = (t >> 3) + 32 * i;
            if (c0 + c < C && m0 + ch * 8 < M) {      // M % 8 == 0 (host contract): a chunk is either fully inside or outside
                H8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o.h[e] = tile[ch * 8 + e][c];
                stg16(dst + (long)(c0 + c) * ldd + m0 + ch * 8, o.u);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ conv dgrad helpers
// Nearest-2x upsampling copies each pixel to a 2x2 block, so its backward is the block sum: dx[b,y,x,:] = sum of the
// four du[b, 2y + i, 2x + j, :] (fp32 in, fp32 out, optional accumulate into out).
__global__ __launch_bounds__(256) void sum2x2_kernel(const float* du, long ldu, float* dx, long ldx, int B, int H, int Wd, int C,
                                                     int accumulate) {
    const int vpr = C / 4;
    const long total = (long)B * H * Wd * vpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % vpr);
        const long pix = i / vpr;
        const int xx = (int)(pix % Wd), yy = (int)((pix / Wd) % H), b = (int)(pix / ((long)Wd * H));
        const long r0 = ((long)b * 2 * H + 2 * yy) * (2 * Wd) + 2 * xx;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = *reinterpret_cast<const float4*>(du + (r0 + (q >> 1) * (2 * Wd) + (q & 1)) * ldu + cv * 4);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        float* o = dx + pix * ldx + cv * 4;
        if (accumulate) {
            const float4 a = *reinterpret_cast<const float4*>(o);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        *reinterpret_cast<float4*>(o) = s;
    }
}

// Stride-2 convolution dgrad = stride-1 convolution (rotated weights) of the gradient scattered onto the even positions
// of a zero image of the INPUT's size.  y is the zero-bordered fp16 conv input [B, 2Ho+2, 2Wo+2, C]: every interior pixel
// is written (zeros at the odd positions), so the buffer needs no clearing between uses.
__global__ __launch_bounds__(256) void zero_stuff_kernel(const void* dy, long lddy, int dy_f32, f16* y, long ldy, int B, int Ho, int Wo,
                                                         int C) {
    const int vpr = C / 8, H = 2 * Ho, Wd = 2 * Wo;
    const long total = (long)B * H * Wd * vpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cv = (int)(i % vpr);
        const long pix = i / vpr;
        const int xx = (int)(pix % Wd), yy = (int)((pix / Wd) % H), b = (int)(pix / ((long)Wd * H));
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (((xx | yy) & 1) == 0) load8f(dy, (((long)b * Ho + (yy >> 1)) * Wo + (xx >> 1)) * lddy + cv * 8, dy_f32, v);
        store8h(y + (((long)b * (H + 2) + yy + 1) * (Wd + 2) + xx + 1) * ldy + cv * 8, v);
    }
}

// ------------------------------------------------------------------------------------------------ loss
// loss = mean(((pred - noise) * keep)^2) (train_StorySalon_stage2.py:325 with keep = 1 - mask), d_pred = 2 (pred - noise)
// keep^2 / n.  One workgroup: n = B*4*h*w is a few 10^4 elements; the sum is a fixed-order tree (deterministic).
__global__ __launch_bounds__(1024) void mse_grad_kernel(const float* pred, const float* noise, const float* mask, float* d_pred,
                                                        float* loss, long n) {
    __shared__ float s_red[16];
    const int t = threadIdx.x;
    float acc = 0.f;
    const float inv_n = 1.0f / (float)n;
    for (long i = t; i < n; i += 1024) {
        const float keep = 1.0f - mask[i];
        const float d = (pred[i] - noise[i]) * keep;
        acc += d * d;
        d_pred[i] = 2.0f * d * keep * inv_n;
    }
    acc = wave_sum(acc);
    if ((t & 63) == 0) s_red[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += s_red[k];
        *loss = s * inv_n;
    }
}

}  // namespace

extern "C" int sg_layernorm_bwd_f16(const void* x, int64_t ldx, int32_t x_f32, const void* dy1, int64_t lddy1,
                                    const sg_half* gamma1, const void* dy2, int64_t lddy2, const sg_half* gamma2, int32_t dy_f32,
                                    const float* res, int64_t ldr, float res_scale, float* out, int64_t ldo, int32_t M,
                                    int32_t C, float eps, sg_stream_t stream) {
    SG_REQUIRE(x && dy1 && gamma1 && out, "sg_layernorm_bwd: null pointer");
    SG_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "sg_layernorm_bwd: C=%d must be a multiple of 8, <= 2048", C);
    SG_REQUIRE(ldx % 8 == 0 && lddy1 % 8 == 0 && ldo % 8 == 0 && ldx >= C && lddy1 >= C && ldo >= C, "sg_layernorm_bwd: bad ld");
    SG_REQUIRE(sg_aligned16(x) && sg_aligned16(dy1) && sg_aligned16(gamma1) && sg_aligned16(out), "sg_layernorm_bwd: 16-byte alignment");
    SG_REQUIRE((dy2 == nullptr) == (gamma2 == nullptr), "sg_layernorm_bwd: dy2 and gamma2 go together");
    SG_REQUIRE(!dy2 || (sg_aligned16(dy2) && sg_aligned16(gamma2) && lddy2 % 8 == 0 && lddy2 >= C), "sg_layernorm_bwd: second pair");
    SG_REQUIRE(!res || (sg_aligned16(res) && ldr % 8 == 0 && ldr >= C), "sg_layernorm_bwd: res alignment / ld");
    LnBwdParams p{};
    p.x = x; p.ldx = ldx; p.x_f32 = x_f32 ? 1 : 0;
    p.dy1 = dy1; p.lddy1 = lddy1; p.g1 = reinterpret_cast<const f16*>(gamma1);
    p.dy2 = dy2; p.lddy2 = lddy2; p.g2 = reinterpret_cast<const f16*>(gamma2);
    p.dy_f32 = dy_f32 ? 1 : 0;
    p.res = res; p.ldr = ldr; p.res_scale = res_scale;
    p.out = out; p.ldo = ldo; p.M = M; p.C = C; p.eps = eps;
    const dim3 grid(sg_cdiv(M, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int nv = sg_cdiv(C / 8, 64);
    if (nv == 1) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, block, 0, st, p);
    else if (nv == 2) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, block, 0, st, p);
    else if (nv == 3) hipLaunchKernelGGL(layernorm_bwd_kernel<3>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, block, 0, st, p);
    SG_CHECK_LAUNCH("sg_layernorm_bwd_f16");
    return SG_OK;
}

extern "C" int sg_geglu_bwd_f16(const sg_half* proj, int64_t ldp, const sg_half* du, int64_t ldu, sg_half* dproj, int64_t lddp,
                                int32_t M, int32_t N8, sg_stream_t stream) {
    SG_REQUIRE(proj && du && dproj, "sg_geglu_bwd: null pointer");
    SG_REQUIRE(M > 0 && N8 > 0 && N8 % 64 == 0, "sg_geglu_bwd: N8=%d must be a multiple of 64 (32 value + 32 gate columns)", N8);
    SG_REQUIRE(ldp % 8 == 0 && ldu % 8 == 0 && lddp % 8 == 0 && ldp >= N8 && lddp >= N8 && ldu >= N8 / 2, "sg_geglu_bwd: bad ld");
    SG_REQUIRE(sg_aligned16(proj) && sg_aligned16(du) && sg_aligned16(dproj), "sg_geglu_bwd: 16-byte alignment");
    const long items = (long)M * (N8 / 16);
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3((int)min((long)4096, (items + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f16*>(proj), (long)ldp, reinterpret_cast<const f16*>(du), (long)ldu,
                       reinterpret_cast<f16*>(dproj), (long)lddp, M, N8);
    SG_CHECK_LAUNCH("sg_geglu_bwd_f16");
    return SG_OK;
}

extern "C" int sg_transpose_f16(const void* src, int64_t lds, int32_t src_f32, sg_half* dst, int64_t ldd, int32_t M, int32_t C,
                                sg_stream_t stream) {
    SG_REQUIRE(src && dst, "sg_transpose: null pointer");
    SG_REQUIRE(M > 0 && C > 0 && M % 8 == 0 && C % 8 == 0, "sg_transpose: M=%d and C=%d must be multiples of 8", M, C);
    SG_REQUIRE(lds % 8 == 0 && ldd % 8 == 0 && lds >= C && ldd >= M, "sg_transpose: bad ld");
    SG_REQUIRE(sg_aligned16(src) && sg_aligned16(dst), "sg_transpose: 16-byte alignment");
    hipLaunchKernelGGL(transpose_kernel, dim3(sg_cdiv(M, 64), sg_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, src, (long)lds, 0L,
                       src_f32 ? 1 : 0, reinterpret_cast<f16*>(dst), (long)ldd, 0L, M, C);
    SG_CHECK_LAUNCH("sg_transpose_f16");
    return SG_OK;
}

extern "C" int sg_transpose_batched_f16(const void* src, int64_t lds, int64_t bs_src, int32_t src_f32, sg_half* dst, int64_t ldd,
                                        int64_t bs_dst, int32_t B, int32_t M, int32_t C, sg_stream_t stream) {
    SG_REQUIRE(src && dst, "sg_transpose_batched: null pointer");
    SG_REQUIRE(B > 0 && B <= 65535 && M > 0 && C > 0 && M % 8 == 0 && C % 8 == 0,
               "sg_transpose_batched: B=%d in 1..65535, M=%d and C=%d must be multiples of 8", B, M, C);
    SG_REQUIRE(lds % 8 == 0 && ldd % 8 == 0 && lds >= C && ldd >= M && bs_src % 8 == 0 && bs_dst % 8 == 0, "sg_transpose_batched: bad ld / batch stride");
    SG_REQUIRE(sg_aligned16(src) && sg_aligned16(dst), "sg_transpose_batched: 16-byte alignment");
    hipLaunchKernelGGL(transpose_kernel, dim3(sg_cdiv(M, 64), sg_cdiv(C, 64), B), dim3(256), 0, (hipStream_t)stream, src, (long)lds,
                       (long)bs_src, src_f32 ? 1 : 0, reinterpret_cast<f16*>(dst), (long)ldd, (long)bs_dst, M, C);
    SG_CHECK_LAUNCH("sg_transpose_batched_f16");
    return SG_OK;
}

extern "C" int sg_sum2x2_f32(const float* du, int64_t ldu, float* dx, int64_t ldx, int32_t B, int32_t H, int32_t W, int32_t C,
                             int32_t accumulate, sg_stream_t stream) {
    SG_REQUIRE(du && dx, "sg_sum2x2: null pointer");
    SG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "sg_sum2x2: bad shape");
    SG_REQUIRE(ldu % 4 == 0 && ldx % 4 == 0 && ldu >= C && ldx >= C && sg_aligned16(du) && sg_aligned16(dx), "sg_sum2x2: ld / alignment");
    const long total = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(sum2x2_kernel, dim3((int)min((long)4096, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, du,
                       (long)ldu, dx, (long)ldx, B, H, W, C, accumulate ? 1 : 0);
    SG_CHECK_LAUNCH("sg_sum2x2_f32");
    return SG_OK;
}

extern "C" int sg_zero_stuff_f16(const void* dy, int64_t lddy, int32_t dy_f32, sg_half* y, int64_t ldy, int32_t B, int32_t Ho,
                                 int32_t Wo, int32_t C, sg_stream_t stream) {
    SG_REQUIRE(dy && y, "sg_zero_stuff: null pointer");
    SG_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "sg_zero_stuff: bad shape");
    SG_REQUIRE(lddy % 8 == 0 && ldy % 8 == 0 && lddy >= C && ldy >= C && sg_aligned16(dy) && sg_aligned16(y), "sg_zero_stuff: ld / alignment");
    const long total = (long)B * 4 * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(zero_stuff_kernel, dim3((int)min((long)4096, (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy,
                       (long)lddy, dy_f32 ? 1 : 0, reinterpret_cast<f16*>(y), (long)ldy, B, Ho, Wo, C);
    SG_CHECK_LAUNCH("sg_zero_stuff_f16");
    return SG_OK;
}

extern "C" int sg_mse_grad_f32(const float* pred, const float* noise, const float* mask, float* d_pred, float* loss, int64_t n,
                               sg_stream_t stream) {
    SG_REQUIRE(pred && noise && mask && d_pred && loss, "sg_mse_grad: null pointer");
    SG_REQUIRE(n > 0, "sg_mse_grad: empty input");
    hipLaunchKernelGGL(mse_grad_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, noise, mask, d_pred, loss, (long)n);
    SG_CHECK_LAUNCH("sg_mse_grad_f32");
    return SG_OK;
}
