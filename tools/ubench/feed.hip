// Microbenchmark: how fast can one CU pull L2-resident data into LDS with LDS-DMA (global_load_lds, 16 B/lane)?
// Each workgroup re-reads its own REGION bytes (L2-resident after the first sweep) into a ring of LDS slots; no compute.
// Build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/feed.hip -o /tmp/feed && /tmp/feed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int NW, int INFLIGHT>
__global__ __launch_bounds__(64 * NW) void feed_kernel(const char* src, long region, int sweeps, int rows_mode) {
    __shared__ __attribute__((aligned(16))) char smem[NW * INFLIGHT * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (long)blockIdx.x * region;
    char* lds = smem + wave * INFLIGHT * 1024;
    const long per_wave = region / NW;                  // >= 8 KiB
    const char* wbase = base + wave * per_wave;
    // rows_mode 0: each wave instruction reads 1 KiB contiguous;
    // rows_mode 1: 8 rows x 128 B, rows 1 KiB apart (the GEMM operand pattern: one cache line per row)
    const long lane_off = rows_mode ? (long)(lane >> 3) * 1024 + (lane & 7) * 16 : (long)lane * 16;
    const int n_instr = (int)(per_wave / 1024);
    for (int s = 0; s < sweeps; ++s) {
        for (int it = 0; it < n_instr; it += INFLIGHT) {
#pragma unroll
            for (int k = 0; k < INFLIGHT; ++k) {
                const int j = (it + k) % n_instr;
                const long off = rows_mode ? (long)(j >> 3) * 8192 + (j & 7) * 128 : (long)j * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + off + lane_off),
                                                 (__attribute__((address_space(3))) void*)(lds + k * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (smem[threadIdx.x] == 123 && sweeps < 0) printf("x");   // keep LDS live
}

template <int NW, int INFLIGHT>
void run(const char* d, long region, int blocks, int rows_mode) {
    const int sweeps = 40;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((feed_kernel<NW, INFLIGHT>), dim3(blocks), dim3(64 * NW), 0, 0, d, region, 2, rows_mode);   // warm L2
    hipEventRecord(a);
    hipLaunchKernelGGL((feed_kernel<NW, INFLIGHT>), dim3(blocks), dim3(64 * NW), 0, 0, d, region, sweeps, rows_mode);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * region * sweeps;
    printf("NW=%2d inflight=%2d rows=%d blocks=%4d region=%4ld KB: %7.2f TB/s total, %6.1f GB/s per WG, %5.1f B/clk/WG @2.1GHz\n", NW, INFLIGHT,
           rows_mode, blocks, region >> 10, bytes / ms / 1e9, bytes / blocks / ms / 1e6, bytes / blocks / (ms * 1e-3) / 2.1e9);
}

int main() {
    const long total = 256L << 20;
    char* d; hipMalloc(&d, total); hipMemset(d, 1, total);
    for (int rows = 0; rows < 2; ++rows) {
        run<4, 4>(d, 128 << 10, 256, rows);
        run<4, 16>(d, 128 << 10, 256, rows);
        run<8, 4>(d, 128 << 10, 256, rows);
        run<8, 8>(d, 128 << 10, 256, rows);
        run<8, 16>(d, 128 << 10, 256, rows);
        run<16, 8>(d, 128 << 10, 256, rows);
        run<8, 8>(d, 64 << 10, 512, rows);      // 2 WGs per CU (LDS 64 KiB each)
        run<8, 8>(d, 1024 << 10, 256, rows);    // 256 MB total: beyond L2 (MALL / HBM)
    }
    return 0;
}
