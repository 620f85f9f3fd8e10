// Issue cost of a few VALU instructions on gfx950, one to four waves per SIMD (development probe; build + run: tools/probes/run_valu_rates.sh).
// Each wave runs REP iterations of 16 independent instructions of one kind and reports cycles (s_memtime) per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 2048

template <int KIND>
__global__ __launch_bounds__(1024) void probe(float* out, long long* cyc, float seed) {
    float r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = seed + 0.01f * (float)(j + threadIdx.x % 7);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[j]));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[j]));
            if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(r[j]));
            if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(r[j]));
            if (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[j]));
            if (KIND == 5) asm volatile("v_exp_f16 %0, %0" : "+v"(r[j]));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += r[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 4 * sizeof(float));
    hipMalloc(&cyc, 64 * sizeof(long long));
    for (int waves_per_simd = 1; waves_per_simd <= 4; ++waves_per_simd) {
        const int threads = 256 * waves_per_simd;          // one workgroup on one CU: 4 SIMDs x waves_per_simd waves
        hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.5f);
        hipDeviceSynchronize();
        std::vector<long long> h(threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        long long mx = 0;
        for (auto v : h) mx = v > mx ? v : mx;
        // s_memtime counts at 100 MHz on gfx9 (constant-rate counter): report both raw ticks and SIMD-cycles at the measured clock ratio
        printf("%-18s waves/SIMD %d: %8.3f ticks per instruction and wave, %8.3f per instruction issued on the SIMD\n", name, waves_per_simd,
               (double)mx / (REP * 16.0), (double)mx / (REP * 16.0 * waves_per_simd));
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1>("v_fma_f32");
    run<0>("v_exp_f32");
    run<5>("v_exp_f16");
    run<4>("v_rcp_f32");
    run<2>("v_max3_f32");
    run<3>("v_cvt_pk_f16_f32");
    return 0;
}
