// Hardware probe (development; not part of the library): what does a layer boundary cost INSIDE one persistent launch on this chip?
//
// SURVEY 8(f) rank 2 / VERDICT r5 item 3 ask for the 8x8 level of the main pass (~30 launches on M = 192 rows) as ONE persistent kernel
// with a grid barrier per layer.  What such a kernel pays per layer instead of a kernel boundary is measured here in isolation: N rounds
// of { every workgroup writes `bytes` of fp32 (a layer's output slab), grid barrier, every workgroup reads what ANOTHER workgroup (on
// another XCD) wrote and checks it } — against the same rounds as N dependent kernel launches in a hipGraph (tools/bench_chain.py reads
// 1.8 us per empty dependent launch on this box).  Two barrier forms: one device-scope counter, and the XCD-hierarchical form (per-XCD
// counter, the XCD's last arriver forwards to a top counter).  Every spin is bounded (a stuck barrier sets a flag and the kernel ends).
//
//   hipcc --offload-arch=gfx950 -O3 -o storygen_amd/lib/probe_grid_barrier tools/probes/grid_barrier.hip && storygen_amd/lib/probe_grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int SPIN_LIMIT = 4000000;

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// form 0: one monotonic counter.  form 1: per-XCD counters (workgroup b runs on XCD b % 8) + a top counter.
template <int FORM>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, int round, int nwg, unsigned* fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FORM == 0) {
            atomicAdd(ctr, 1u);
            const unsigned want = (unsigned)(round + 1) * (unsigned)nwg;
            int spins = 0;
            while (ld_relaxed(ctr) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { atomicExch(fail, 1u); ok = false; break; }
            }
        } else {
            const int xcd = blockIdx.x & 7, per = (nwg + 7 - xcd) / 8;          // workgroups on this XCD
            const unsigned old = atomicAdd(ctr + 16 * (1 + xcd), 1u);           // (counters 64 bytes apart)
            if (old == (unsigned)(round + 1) * (unsigned)per - 1u) atomicAdd(ctr, 1u);      // the XCD's last arriver
            const unsigned want = (unsigned)(round + 1) * 8u;
            int spins = 0;
            while (ld_relaxed(ctr) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) { atomicExch(fail, 1u); ok = false; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

template <int FORM>
__global__ __launch_bounds__(256) void persistent(float* buf, int floats_per_wg, int rounds, unsigned* ctr, unsigned* fail, unsigned* bad) {
    const int nwg = gridDim.x;
    unsigned mism = 0;
    for (int r = 0; r < rounds; ++r) {
        // double-buffered slabs: ONE barrier per round (a slab is overwritten two rounds later, behind the next barrier)
        float* mine = buf + (size_t)blockIdx.x * floats_per_wg + (size_t)(r & 1) * nwg * floats_per_wg;
        for (int i = threadIdx.x * 4; i < floats_per_wg; i += 256 * 4)
            *reinterpret_cast<float4*>(mine + i) = make_float4((float)r, (float)r, (float)r, (float)r);
        if (!grid_barrier<FORM>(ctr, r, nwg, fail)) return;
        const float* other = buf + (size_t)((blockIdx.x + 3) % nwg) * floats_per_wg + (size_t)(r & 1) * nwg * floats_per_wg;   // a neighbour on another XCD
        for (int i = threadIdx.x * 4; i < floats_per_wg; i += 256 * 4) {
            const float4 v = *reinterpret_cast<const float4*>(other + i);
            mism += (v.x != (float)r) + (v.w != (float)r);
        }
    }
    if (mism) atomicAdd(bad, mism);
}

__global__ __launch_bounds__(256) void one_round(float* buf, int floats_per_wg, int r, unsigned* bad) {
    const int nwg = gridDim.x;
    unsigned mism = 0;
    if (r > 0) {           // read what the PREVIOUS launch wrote (the dependency a layer has on its predecessor)
        const float* other = buf + (size_t)((blockIdx.x + 3) % nwg) * floats_per_wg + (size_t)((r - 1) & 1) * nwg * floats_per_wg;
        for (int i = threadIdx.x * 4; i < floats_per_wg; i += 256 * 4) {
            const float4 v = *reinterpret_cast<const float4*>(other + i);
            mism += (v.x != (float)(r - 1)) + (v.w != (float)(r - 1));
        }
    }
    float* mine = buf + (size_t)blockIdx.x * floats_per_wg + (size_t)(r & 1) * nwg * floats_per_wg;
    for (int i = threadIdx.x * 4; i < floats_per_wg; i += 256 * 4)
        *reinterpret_cast<float4*>(mine + i) = make_float4((float)r, (float)r, (float)r, (float)r);
    if (mism) atomicAdd(bad, mism);
}

int main() {
    const int rounds = 200;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned *ctr, *fail, *bad;
    CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&bad, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("%-10s %-12s %14s %14s %16s\n", "workgroups", "KB per WG", "1 counter us", "XCD form us", "graph launches us");
    for (int nwg : {240, 256}) {
        for (int kb : {0, 4, 16, 64}) {
            const int fl = kb ? kb * 256 : 1024;           // (0 KB: a 4 KB slab that is neither written nor read is not possible here: use 4 KB and report as 0 -> skip)
            if (kb == 0) continue;
            float* buf;
            CK(hipMalloc(&buf, (size_t)2 * nwg * fl * 4));
            float us[3] = {0, 0, 0};
            unsigned h_fail = 0, h_bad = 0;
            for (int form = 0; form < 2; ++form) {
                CK(hipMemsetAsync(ctr, 0, 4096, st)); CK(hipMemsetAsync(fail, 0, 4, st)); CK(hipMemsetAsync(bad, 0, 4, st));
                // warm-up + timed
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemsetAsync(ctr, 0, 4096, st));
                    CK(hipEventRecord(a, st));
                    if (form == 0) hipLaunchKernelGGL(persistent<0>, dim3(nwg), dim3(256), 0, st, buf, fl, rounds, ctr, fail, bad);
                    else hipLaunchKernelGGL(persistent<1>, dim3(nwg), dim3(256), 0, st, buf, fl, rounds, ctr, fail, bad);
                    CK(hipEventRecord(b, st));
                    CK(hipStreamSynchronize(st));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    us[form] = ms * 1e3f / rounds;          // one barrier per round
                }
                unsigned f, bd;
                CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&bd, bad, 4, hipMemcpyDeviceToHost));
                h_fail |= f; h_bad += bd;
            }
            // the same rounds as dependent launches of one captured graph (double-buffered slabs: one boundary per round)
            CK(hipMemsetAsync(bad, 0, 4, st));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(one_round, dim3(nwg), dim3(256), 0, st, buf, fl, r, bad);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(a, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(b, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            us[2] = ms * 1e3f / rounds;
            unsigned bd; CK(hipMemcpy(&bd, bad, 4, hipMemcpyDeviceToHost));
            printf("%-10d %-12d %14.2f %14.2f %16.2f   %s%s\n", nwg, kb, us[0], us[1], us[2], h_fail ? "BARRIER TIMED OUT " : "",
                   (h_bad || bd) ? "STALE DATA SEEN" : "all reads current");
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            CK(hipFree(buf));
        }
    }
    printf("(persistent columns: us per round = write slab + barrier + read a neighbour's slab; graph column: us per launch = read previous + write)\n");
    return 0;
}
