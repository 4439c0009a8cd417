// Experiment (round 6): what a device-wide barrier inside a persistent kernel costs on MI355X, against the boundary between two
// dependent kernels of one stream -- the price list for "one persistent binning kernel with grid barriers between its passes".
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier scripts/xp/grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef SLEEP
#define SLEEP 2
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

// flat: one counter, one generation word (sense by generation number)
__global__ void flat_barriers(unsigned* ctr, unsigned* gen, int n_bar, unsigned* sink)
{
    unsigned acc = 0;
    for (int b = 0; b < n_bar; ++b) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned arrived = atomicAdd(ctr, 1u) + 1u;
            if (arrived == (unsigned)(b + 1) * gridDim.x) __hip_atomic_store(gen, (unsigned)(b + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else {
                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(b + 1)) __builtin_amdgcn_s_sleep(SLEEP);
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
            }
        }
        __syncthreads();
        acc += b;
    }
    if (acc == 0xffffffffu) *sink = acc;
}

// hierarchical: a counter per XCD (its L2), the last block of an XCD reports to the global counter, every XCD spins on its OWN flag word
__global__ void xcd_barriers(unsigned* xctr /* [8*32] */, unsigned* gctr, unsigned* xgen /* [8*32] */, const unsigned* per_xcd, int n_bar, unsigned* sink)
{
    const unsigned x = xcc_id();
    unsigned acc = 0;
    for (int b = 0; b < n_bar; ++b) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned a = atomicAdd(xctr + 32 * x, 1u) + 1u;
            if (a == (unsigned)(b + 1) * per_xcd[x]) {
                const unsigned g = atomicAdd(gctr, 1u) + 1u;
                if (g == (unsigned)(b + 1) * 8u)
                    for (int k = 0; k < 8; ++k) __hip_atomic_store(xgen + 32 * k, (unsigned)(b + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            while (__hip_atomic_load(xgen + 32 * x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(b + 1)) __builtin_amdgcn_s_sleep(SLEEP);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        acc += b;
    }
    if (acc == 0xffffffffu) *sink = acc;
}

__global__ void count_xcd(unsigned* per_xcd) { if (threadIdx.x == 0) atomicAdd(per_xcd + xcc_id(), 1u); }
__global__ void tiny(unsigned* p, int k) { if (p[0] == 0xdeadbeefu) p[1] = k; }

int main()
{
    unsigned* d; CK(hipMalloc(&d, 1 << 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n_bar = 200;
    for (int blocks : { 256, 512, 1024, 2048 }) {
        for (int threads : { 256, 1024 }) {
            if (blocks * threads > 256 * 2048) continue;                     // must all be resident
            float best_flat = 1e9f, best_x = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(d, 0, 1 << 16));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(flat_barriers, dim3(blocks), dim3(threads), 0, 0, d, d + 64, n_bar, d + 128);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_flat) best_flat = ms;
                CK(hipMemset(d, 0, 1 << 16));
                hipLaunchKernelGGL(count_xcd, dim3(blocks), dim3(threads), 0, 0, d + 4096);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(xcd_barriers, dim3(blocks), dim3(threads), 0, 0, d, d + 1024, d + 2048, d + 4096, n_bar, d + 128);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_x) best_x = ms;
            }
            printf("grid barrier, %4d blocks x %4d threads: flat %.2f us, per-XCD counters %.2f us each (%d barriers in one launch)\n", blocks, threads,
                   1e3f * best_flat / n_bar, 1e3f * best_x / n_bar, n_bar);
        }
    }
    // the boundary between dependent kernels of one stream: 200 tiny kernels back to back
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, 0, d, k);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("200 dependent tiny kernels (256 x 256) in one stream: %.2f us each (launch-rate bound when the host is the limit)\n", 1e3f * ms / 200);
    }
    // the same as a graph (no host in the loop)
    {
        hipStream_t s; CK(hipStreamCreate(&s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, s, d, k);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("the same 200 kernels as one graph launch: %.2f us each\n", 1e3f * ms / 200);
        }
    }
    return 0;
}
