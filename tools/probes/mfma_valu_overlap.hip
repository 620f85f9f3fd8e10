// Do the matrix pipe and the VALU of one SIMD overlap across waves on gfx950?  (development probe; tools/probes/README.md)
// One workgroup = 8 waves = 2 per SIMD.  Waves 0-3 (one per SIMD) run ROLE_A, waves 4-7 run ROLE_B; each wave times its own loop with
// s_memtime.  Roles: 0 idle, 1 = independent v_mfma_f32_32x32x16_f16 (4 accumulator sets in VGPRs), 2 = v_exp_f32, 3 = v_max3_f32,
// 4 = the same MFMAs as one dependent chain (one accumulator set), 5 = v_cvt_pk_f16_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP 1024

__device__ __forceinline__ float role_mfma(int sets, float seed) {
    f16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + j); b[j] = (_Float16)(seed - j); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < REP; ++i) {
        if (sets == 4) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        }
    }
    return c0[0] + c1[1] + c2[2] + c3[3];
}

template <int KIND>
__device__ __forceinline__ float role_valu(float seed) {
    float r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = seed + 0.01f * (float)j;
    for (int i = 0; i < REP / 2; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(r[j]));
            if (KIND == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(r[j]));
            if (KIND == 5) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(r[j]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += r[j];
    return s;
}

__device__ __forceinline__ float run_role(int role, float seed) {
    switch (role) {
        case 1: return role_mfma(4, seed);
        case 4: return role_mfma(1, seed);
        case 2: return role_valu<2>(seed);
        case 3: return role_valu<3>(seed);
        case 5: return role_valu<5>(seed);
        default: return 0.f;
    }
}

__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int role_a, int role_b, float seed) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_a : role_b;        // uniform per wave
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const float v = run_role(__builtin_amdgcn_readfirstlane(role), seed);
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = v;
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 512 * sizeof(float));
    hipMalloc(&cyc, 8 * sizeof(long long));
    const char* names[] = {"idle", "mfma x4 indep", "v_exp_f32", "v_max3_f32", "mfma chain", "v_cvt_pk"};
    const int ops[] = {0, REP * 4, REP * 8, REP * 8, REP * 4, REP * 8};
    const int pairs[][2] = {{1, 0}, {4, 0}, {0, 2}, {0, 3}, {0, 5}, {1, 2}, {1, 3}, {1, 5}, {4, 2}, {4, 3}, {1, 1}, {2, 2}, {3, 3}, {2, 3}};
    for (auto& p : pairs) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, out, cyc, p[0], p[1], 0.5f);
        hipDeviceSynchronize();
        long long h[8];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        long long ma = 0, mb = 0;
        for (int w = 0; w < 4; ++w) { ma = h[w] > ma ? h[w] : ma; mb = h[4 + w] > mb ? h[4 + w] : mb; }
        printf("A = %-14s B = %-14s | A: %8.2f ticks/op   B: %8.2f ticks/op\n", names[p[0]], names[p[1]],
               p[0] ? (double)ma / ops[p[0]] : 0.0, p[1] ? (double)mb / ops[p[1]] : 0.0);
    }
    return 0;
}
