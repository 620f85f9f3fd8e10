// Optimizer step of the stage-2 / COCO training loop (/root/reference/train_StorySalon_stage2.py:186-205,328-333:
// accelerator.clip_grad_norm_ -> optimizer.step() with torch.optim.AdamW or bitsandbytes' AdamW8bit).  HBM-bound elementwise kernels:
//   sumsq_kernel / sumsq_final_kernel   deterministic sum of squares of one gradient tensor (two levels, no atomics)
//   adamw_kernel                        torch.optim.AdamW's update on fp32 states, gradient unscale + global-norm clipping fused in
//   adamw8_kernel                       the same update on block-wise 8-bit states (2048-element blocks, dynamic-tree code books,
//                                       per-block absmax) — the published algorithm of bitsandbytes' 8-bit optimizers, which are CUDA-only
// The clipping coefficient is computed on the device from the per-tensor sums of squares, so a whole optimizer step needs no host read.
#include "common.h"

namespace {

constexpr int SS_THREADS = 256, SS_MAX_BLOCKS = 1024;

__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float block_max256(float v, float* red) {
    v = wave_max_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(SS_THREADS) void sumsq_kernel(const float* x, long n, float* partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * SS_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * SS_THREADS) s += x[i] * x[i];
    s = block_sum256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(SS_THREADS) void sumsq_final_kernel(const float* partial, int nparts, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += SS_THREADS) s += partial[i];
    s = block_sum256(s, red);
    if (threadIdx.x == 0) *out = s;
}

struct AdamParams {
    float* p; const float* g; long n;
    float lr, beta1, beta2, eps, wd, bc1, sbc2;      // bc1 = 1 - beta1^t, sbc2 = sqrt(1 - beta2^t)
    float gscale;                                    // gradients are multiplied by this (1 / loss scale)
    const float* sumsq; int nsumsq; float max_norm;  // clip: g *= min(1, max_norm / (gscale * sqrt(sum sumsq) + 1e-6)); NULL = none
};

__device__ __forceinline__ float clip_factor(const AdamParams& a) {
    if (!a.sumsq) return a.gscale;
    float t = 0.f;
    for (int i = 0; i < a.nsumsq; ++i) t += a.sumsq[i];          // same order in every thread: one value for the whole step
    const float norm = a.gscale * sqrtf(t);
    return a.gscale * fminf(1.0f, a.max_norm / (norm + 1e-6f));
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamParams a, float* m, float* v) {
    const float gs = clip_factor(a);
    const float step = a.lr / a.bc1;                                // torch.optim.AdamW's order of operations (_single_tensor_adamw)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
        const float g = a.g[i] * gs;
        const float mi = m[i] + (g - m[i]) * (1.0f - a.beta1);       // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = a.beta2 * v[i] + (1.0f - a.beta2) * g * g;
        m[i] = mi; v[i] = vi;
        const float p = a.p[i] * (1.0f - a.lr * a.wd);
        a.p[i] = p - step * (mi / (sqrtf(vi) / a.sbc2 + a.eps));
    }
}

constexpr int Q_BLOCK = 2048;

// index of the code-book entry nearest to x (code sorted ascending, 256 entries; ties go to the lower index)
__device__ __forceinline__ int nearest_code(const float* code, float x) {
    int lo = 0, hi = 255;                     // invariant: code[lo] <= x < code[hi] after clamping
    if (x <= code[0]) return 0;
    if (x >= code[255]) return 255;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int mid = (lo + hi) >> 1;
        if (code[mid] <= x) lo = mid; else hi = mid;
    }
    return (x - code[lo] <= code[hi] - x) ? lo : hi;
}

__global__ __launch_bounds__(256) void adamw8_kernel(const AdamParams a, unsigned char* c1, unsigned char* c2, float* amax1, float* amax2,
                                                     const float* code1, const float* code2) {
    __shared__ float s1[256], s2[256], red[4];
    const int t = threadIdx.x;
    s1[t] = code1[t]; s2[t] = code2[t];
    __syncthreads();
    const float gs = clip_factor(a);
    const float step = a.lr * a.sbc2 / a.bc1;
    const long base = (long)blockIdx.x * Q_BLOCK;
    const float a1 = amax1[blockIdx.x], a2 = amax2[blockIdx.x];
    float m[8], v[8];
    float mx1 = 0.f, mx2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long i = base + j * 256 + t;
        m[j] = v[j] = 0.f;
        if (i < a.n) {
            const float g = a.g[i] * gs;
            m[j] = a.beta1 * (s1[c1[i]] * a1) + (1.0f - a.beta1) * g;
            v[j] = a.beta2 * (s2[c2[i]] * a2) + (1.0f - a.beta2) * g * g;
            mx1 = fmaxf(mx1, fabsf(m[j]));
            mx2 = fmaxf(mx2, v[j]);
            float p = a.p[i];
            p -= a.lr * a.wd * p;
            a.p[i] = p - step * (m[j] / (sqrtf(v[j]) + a.eps * a.sbc2));
        }
    }
    mx1 = block_max256(mx1, red);
    mx2 = block_max256(mx2, red);
    if (t == 0) { amax1[blockIdx.x] = mx1; amax2[blockIdx.x] = mx2; }
    const float r1 = mx1 > 0.f ? 1.0f / mx1 : 0.f, r2 = mx2 > 0.f ? 1.0f / mx2 : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const long i = base + j * 256 + t;
        if (i < a.n) {
            c1[i] = (unsigned char)nearest_code(s1, m[j] * r1);
            c2[i] = (unsigned char)nearest_code(s2, v[j] * r2);
        }
    }
}

int fill(AdamParams& a, const sg_adamw_desc* d, const char* who) {
    SG_REQUIRE(d != nullptr, "%s: null descriptor", who);
    SG_REQUIRE(d->param && d->grad && d->n > 0, "%s: null param / grad or empty tensor", who);
    SG_REQUIRE(d->step >= 1, "%s: step counts from 1 (got %d)", who, d->step);
    SG_REQUIRE(d->lr >= 0.f && d->beta1 >= 0.f && d->beta1 < 1.f && d->beta2 >= 0.f && d->beta2 < 1.f && d->eps >= 0.f && d->weight_decay >= 0.f,
               "%s: bad hyper-parameters", who);
    SG_REQUIRE(!d->sumsq || (d->n_sumsq > 0 && d->n_sumsq <= 4096 && d->max_norm > 0.f), "%s: clipping needs 1..4096 sums of squares and max_norm > 0", who);
    a.p = d->param; a.g = d->grad; a.n = d->n;
    a.lr = d->lr; a.beta1 = d->beta1; a.beta2 = d->beta2; a.eps = d->eps; a.wd = d->weight_decay;
    a.bc1 = (float)(1.0 - pow((double)d->beta1, (double)d->step));
    a.sbc2 = (float)sqrt(1.0 - pow((double)d->beta2, (double)d->step));
    a.gscale = d->grad_scale;
    a.sumsq = d->sumsq; a.nsumsq = d->n_sumsq; a.max_norm = d->max_norm;
    return SG_OK;
}

}  // namespace

extern "C" size_t sg_sumsq_scratch_floats(void) { return SS_MAX_BLOCKS; }

extern "C" int sg_sumsq_f32(const float* x, int64_t n, float* out, float* scratch, sg_stream_t stream) {
    SG_REQUIRE(x && out && scratch && n > 0, "sg_sumsq: null pointer or empty tensor");
    const int blocks = (int)(n / (SS_THREADS * 8) + 1 < SS_MAX_BLOCKS ? n / (SS_THREADS * 8) + 1 : SS_MAX_BLOCKS);
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(SS_THREADS), 0, (hipStream_t)stream, x, (long)n, scratch);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(SS_THREADS), 0, (hipStream_t)stream, scratch, blocks, out);
    SG_CHECK_LAUNCH("sg_sumsq_f32");
    return SG_OK;
}

extern "C" int sg_adamw_f32(const sg_adamw_desc* d, sg_stream_t stream) {
    AdamParams a;
    if (int rc = fill(a, d, "sg_adamw_f32")) return rc;
    SG_REQUIRE(d->exp_avg && d->exp_avg_sq, "sg_adamw_f32: null state");
    const int blocks = (int)(a.n / 1024 + 1 < 4096 ? a.n / 1024 + 1 : 4096);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, d->exp_avg, d->exp_avg_sq);
    SG_CHECK_LAUNCH("sg_adamw_f32");
    return SG_OK;
}

extern "C" size_t sg_adamw8bit_blocks(int64_t n) { return (size_t)((n + Q_BLOCK - 1) / Q_BLOCK); }

extern "C" int sg_adamw8bit(const sg_adamw_desc* d, sg_stream_t stream) {
    AdamParams a;
    if (int rc = fill(a, d, "sg_adamw8bit")) return rc;
    SG_REQUIRE(d->code1 && d->code2 && d->absmax1 && d->absmax2 && d->q_code1 && d->q_code2, "sg_adamw8bit: null 8-bit state / code book");
    hipLaunchKernelGGL(adamw8_kernel, dim3((unsigned)sg_adamw8bit_blocks(a.n)), dim3(256), 0, (hipStream_t)stream, a, d->code1, d->code2,
                       d->absmax1, d->absmax2, d->q_code1, d->q_code2);
    SG_CHECK_LAUNCH("sg_adamw8bit");
    return SG_OK;
}
