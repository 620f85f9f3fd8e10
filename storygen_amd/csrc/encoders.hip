// Small kernels of the two networks either side of the denoising loop: the CLIP text encoder
// (/root/reference/model/pipeline.py:137,183) and the AutoencoderKL (:198-205,392,401).  Their GEMMs, convolutions, GroupNorms and
// LayerNorms are the UNet's kernels (gemm_conv.hip, norm.hip); what is left is bandwidth- or latency-bound glue:
//   softmax_rows_kernel   the VAE mid-block AttentionBlock (ONE head of 512 channels: QK^T and PV are plain GEMMs, fp32 scores)
//   attn_small_kernel     CLIP's causal self-attention (77 tokens, 12 heads of 64): K/V of one (batch, head) live in LDS
//   act_rows_kernel       quick_gelu / gelu between CLIP's fc1 and fc2
//   embed_tokens_kernel   token + position embedding gather into the fp32 residual stream
//   gaussian_sample_kernel  DiagonalGaussianDistribution.sample() * scaling factor
#include "common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// p[row, j] = softmax_j(scale * s[row, j]), j < N; columns [N, Npad) are written as zeros (the PV GEMM's K dimension is Npad).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, long lds, f16* p, long ldp, int N, int Npad, float scale) {
    __shared__ float red[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float* sr = s + (long)blockIdx.x * lds;
    f16* pr = p + (long)blockIdx.x * ldp;
    float m = -INFINITY;
    for (int j = t; j < N; j += 256) m = fmaxf(m, sr[j]);
    m = wave_max(m);
    if (lane == 0) red[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = t; j < N; j += 256) sum += expf(scale * (sr[j] - m));
    sum = wave_sum(sum);
    if (lane == 0) red[w] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / sum;
    for (int j = t; j < Npad; j += 256) pr[j] = j < N ? (f16)(expf(scale * (sr[j] - m)) * inv) : (f16)0.f;
}

constexpr int AS_MAXT = 128, AS_MAXD = 64, AS_LD = AS_MAXD + 2;   // +2 halfs: a row is 33 dwords, lane-per-row reads spread over banks

constexpr int AS_ROWS = 16;     // query rows per workgroup (4 per wave): CLIP's 77 tokens x 12 heads x B spread over 60 B workgroups

// grid (heads, batches, row chunks); one wave per query row at a time, lane j owns keys j and j + 64.  A causal chunk only needs the
// keys up to its last row.
__global__ __launch_bounds__(256) void attn_small_kernel(const f16* q, long ldq, long bsq, const f16* k, long ldk, long bsk,
                                                         const f16* v, long ldv, long bsv, f16* o, long ldo, long bso,
                                                         const float* key_bias, int T, int D, float scale, int causal) {
    __shared__ f16 sK[AS_MAXT][AS_LD], sV[AS_MAXT][AS_LD];
    __shared__ float sQ[4][AS_MAXD], sP[4][AS_MAXT];
    const int h = blockIdx.x, b = blockIdx.y, r0 = blockIdx.z * AS_ROWS, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int Tk = causal ? min(T, r0 + AS_ROWS) : T;             // keys this chunk can see
    const f16* kb = k + (long)b * bsk + h * D;
    const f16* vb = v + (long)b * bsv + h * D;
    for (int idx = t; idx < Tk * D; idx += 256) {
        const int j = idx / D, d = idx - j * D;
        sK[j][d] = kb[(long)j * ldk + d];
        sV[j][d] = vb[(long)j * ldv + d];
    }
    __syncthreads();
    for (int r = 0; r < AS_ROWS / 4; ++r) {
        const int i = r0 + r * 4 + w;
        const bool live = i < T;
        if (live && lane < D) sQ[w][lane] = (float)q[(long)b * bsq + (long)i * ldq + h * D + lane] * scale;
        __syncthreads();
        float s0 = -INFINITY, s1 = -INFINITY;
        if (live) {
            const int j0 = lane, j1 = lane + 64;
            if (j0 < Tk && !(causal && j0 > i)) {
                float a = 0.f;
                for (int d = 0; d < D; ++d) a += sQ[w][d] * (float)sK[j0][d];
                s0 = a + (key_bias ? key_bias[(long)b * T + j0] : 0.f);
            }
            if (j1 < Tk && !(causal && j1 > i)) {
                float a = 0.f;
                for (int d = 0; d < D; ++d) a += sQ[w][d] * (float)sK[j1][d];
                s1 = a + (key_bias ? key_bias[(long)b * T + j1] : 0.f);
            }
        }
        const float m = wave_max(fmaxf(s0, s1));
        const float e0 = s0 == -INFINITY ? 0.f : expf(s0 - m), e1 = s1 == -INFINITY ? 0.f : expf(s1 - m);
        const float sum = wave_sum(e0 + e1);
        sP[w][lane] = e0;
        sP[w][lane + 64] = e1;
        __syncthreads();
        if (live && lane < D) {
            const int jn = causal ? min(Tk, i + 1) : Tk;
            float acc = 0.f;
            for (int j = 0; j < jn; ++j) acc += sP[w][j] * (float)sV[j][lane];
            o[(long)b * bso + (long)i * ldo + h * D + lane] = (f16)(acc / sum);
        }
        __syncthreads();
    }
}

// in place: x = x * sigmoid(1.702 x) (act 0, CLIP "quick_gelu") or the erf GELU (act 1)
__global__ __launch_bounds__(256) void act_rows_kernel(f16* x, long ldx, int M, int N, int act) {
    const int vpr = N / 8;
    const long total = (long)M * vpr;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / vpr;
        const int c = (int)(idx - r * vpr);
        f16* px = x + r * ldx + c * 8;
        H8 v; v.u = ldg16(px);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = (float)v.h[j];
            o[j] = act == 0 ? a / (1.0f + expf(-1.702f * a)) : gelu_erf_f(a);
        }
        store8h(px, o);
    }
}

// out[r, :] = tok[ids[r], :] + pos[r % T, :]   (fp32 tables, fp32 residual stream)
__global__ __launch_bounds__(256) void embed_tokens_kernel(const long long* ids, const float* tok, const float* pos, float* out, long ldo,
                                                           int rows, int T, int C) {
    const int vpr = C / 4;
    const long total = (long)rows * vpr;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / vpr;
        const int c = (int)(idx - r * vpr) * 4;
        const float4 a = *reinterpret_cast<const float4*>(tok + (long)ids[r] * C + c);
        const float4 p = *reinterpret_cast<const float4*>(pos + (long)(r % T) * C + c);
        *reinterpret_cast<float4*>(out + r * ldo + c) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float* mean, const float* logvar, const float* noise, float* out,
                                                              float scale, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = mean[i];
        if (noise) v += expf(0.5f * fminf(fmaxf(logvar[i], -30.0f), 20.0f)) * noise[i];
        out[i] = v * scale;
    }
}

}  // namespace

extern "C" int sg_softmax_rows_f16(const float* s, int64_t lds, sg_half* p, int64_t ldp, int32_t M, int32_t N, float scale,
                                   sg_stream_t stream) {
    SG_REQUIRE(s && p, "sg_softmax_rows: null pointer");
    const int Npad = (N + 7) & ~7;
    SG_REQUIRE(M > 0 && N > 0 && lds >= N && ldp >= Npad, "sg_softmax_rows: bad shape M=%d N=%d lds=%lld ldp=%lld", M, N, (long long)lds,
               (long long)ldp);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, s, (long)lds, reinterpret_cast<f16*>(p), (long)ldp, N,
                       Npad, scale);
    SG_CHECK_LAUNCH("sg_softmax_rows_f16");
    return SG_OK;
}

extern "C" int sg_attn_small_f16(const sg_half* q, int64_t ldq, int64_t bsq, const sg_half* k, int64_t ldk, int64_t bsk, const sg_half* v,
                                 int64_t ldv, int64_t bsv, sg_half* o, int64_t ldo, int64_t bso, const float* key_bias, int32_t B,
                                 int32_t H, int32_t T, int32_t D, float scale, int32_t causal, sg_stream_t stream) {
    SG_REQUIRE(q && k && v && o, "sg_attn_small: null pointer");
    SG_REQUIRE(B > 0 && H > 0 && T > 0 && T <= AS_MAXT && D > 0 && D <= AS_MAXD, "sg_attn_small: needs T <= %d and D <= %d (got T=%d D=%d)",
               AS_MAXT, AS_MAXD, T, D);
    SG_REQUIRE(ldq >= (int64_t)H * D && ldk >= (int64_t)H * D && ldv >= (int64_t)H * D && ldo >= (int64_t)H * D, "sg_attn_small: token stride below H*D");
    SG_REQUIRE(causal == 0 || causal == 1, "sg_attn_small: causal must be 0 or 1");
    hipLaunchKernelGGL(attn_small_kernel, dim3(H, B, (T + AS_ROWS - 1) / AS_ROWS), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f16*>(q), (long)ldq, (long)bsq,
                       reinterpret_cast<const f16*>(k), (long)ldk, (long)bsk, reinterpret_cast<const f16*>(v), (long)ldv, (long)bsv,
                       reinterpret_cast<f16*>(o), (long)ldo, (long)bso, key_bias, T, D, scale, causal);
    SG_CHECK_LAUNCH("sg_attn_small_f16");
    return SG_OK;
}

extern "C" int sg_act_rows_f16(sg_half* x, int64_t ldx, int32_t M, int32_t N, int32_t act, sg_stream_t stream) {
    SG_REQUIRE(x, "sg_act_rows: null pointer");
    SG_REQUIRE(M > 0 && N > 0 && N % 8 == 0 && ldx % 8 == 0 && ldx >= N && sg_aligned16(x), "sg_act_rows: N and ldx must be multiples of 8, x 16-byte aligned");
    SG_REQUIRE(act == SG_ACT_QUICK_GELU || act == SG_ACT_GELU, "sg_act_rows: unknown activation %d", act);
    const long total = (long)M * (N / 8);
    const int blocks = (int)(total / 256 + 1 < 65536 ? total / 256 + 1 : 65536);
    hipLaunchKernelGGL(act_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<f16*>(x), (long)ldx, M, N, act);
    SG_CHECK_LAUNCH("sg_act_rows_f16");
    return SG_OK;
}

extern "C" int sg_embed_tokens_f32(const int64_t* ids, const float* tok, const float* pos, float* out, int64_t ldo, int32_t rows, int32_t T,
                                   int32_t C, sg_stream_t stream) {
    SG_REQUIRE(ids && tok && pos && out, "sg_embed_tokens: null pointer");
    SG_REQUIRE(rows > 0 && T > 0 && C > 0 && C % 4 == 0 && ldo % 4 == 0 && ldo >= C, "sg_embed_tokens: bad shape rows=%d T=%d C=%d", rows, T, C);
    SG_REQUIRE(sg_aligned16(tok) && sg_aligned16(pos) && sg_aligned16(out), "sg_embed_tokens: 16-byte alignment");
    const long total = (long)rows * (C / 4);
    const int blocks = (int)(total / 256 + 1 < 65536 ? total / 256 + 1 : 65536);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const long long*>(ids), tok, pos, out,
                       (long)ldo, rows, T, C);
    SG_CHECK_LAUNCH("sg_embed_tokens_f32");
    return SG_OK;
}

extern "C" int sg_gaussian_sample_f32(const float* mean, const float* logvar, const float* noise, float* out, float scale, int64_t n,
                                      sg_stream_t stream) {
    SG_REQUIRE(mean && out && (noise == nullptr || logvar != nullptr), "sg_gaussian_sample: null pointer");
    SG_REQUIRE(n > 0, "sg_gaussian_sample: empty");
    const int blocks = (int)(n / 256 + 1 < 4096 ? n / 256 + 1 : 4096);
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mean, logvar, noise, out, scale, (long)n);
    SG_CHECK_LAUNCH("sg_gaussian_sample_f32");
    return SG_OK;
}
