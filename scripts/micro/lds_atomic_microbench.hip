// Microbenchmark: throughput of LDS atomics by type on MI355X (one 1024-thread workgroup per CU, 128 KB of LDS,
// pseudo-random addresses, ~1/16 of the lanes active per instruction like the hash-grid slab kernel, or all lanes).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o lds_atomic_microbench lds_atomic_microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE, int SPARSE>
__global__ void __launch_bounds__(1024) k(int iters, float* out)
{
    extern __shared__ unsigned char raw[];
    float* f = (float*)raw; unsigned* u = (unsigned*)raw; unsigned long long* q = (unsigned long long*)raw; double* d = (double*)raw;
    for (int i = threadIdx.x; i < 32768; i += 1024) f[i] = 0.0f;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned r = (s >> 8);
        const bool act = SPARSE ? ((r & 15u) == 0u) : true;
        unsigned a = (r >> 4) & 16383u;                          // 16384 slots (x 8 bytes = 128 KB for the 64-bit case)
        if (MODE >= 6) a = __shfl(a, threadIdx.x & ~7u, 64);     // clustered: groups of 8 adjacent lanes hit ONE address (compositor case)
        if (act) {
            if (MODE == 0) atomicAdd(&f[a], 1.0f);
            else if (MODE == 1) atomicAdd(&u[a], 1u);
            else if (MODE == 2) atomicAdd(&q[a], 1ull);
            else if (MODE == 3) f[a] += 1.0f;                    // racy read-modify-write
            else if (MODE == 4) { if (r == 0x7fffffffu) f[a] = 1.0f; }   // no LDS traffic: loop + rng only
            else if (MODE == 5) atomicAdd(&d[a], 1.0);          // ds_add_f64
            else if (MODE == 6) atomicAdd(&q[a], 1ull);
            else if (MODE == 7) atomicAdd(&f[a], 1.0f);
            else if (MODE == 8) atomicAdd(&d[a], 1.0);
        }
    }
    __syncthreads();
    if (f[threadIdx.x] == 123.0f) out[0] = 1.0f;
}

template <int MODE, int SPARSE>
static void run(const char* name)
{
    float* out; hipMalloc(&out, 4);
    const int iters = 4096, blocks = 256;
    hipFuncSetAttribute((const void*)k<MODE, SPARSE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, SPARSE><<<blocks, 1024, 131072>>>(iters, out); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, SPARSE><<<blocks, 1024, 131072>>>(iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * 1024 * iters / (SPARSE ? 16 : 1);
    printf("%-26s %-7s %.3f ms  %.1f G lane-ops/s  (%.2f cycles per lane-op per CU at 2.4 GHz)\n", name, SPARSE ? "sparse" : "dense",
           ms, lane_ops / ms / 1e6, ms * 1e-3 * 2.4e9 / (lane_ops / blocks));
    hipFree(out);
}

int main()
{
    run<4, 1>("loop only");       run<4, 0>("loop only");
    run<0, 1>("ds_add_f32");      run<0, 0>("ds_add_f32");
    run<1, 1>("ds_add_u32");      run<1, 0>("ds_add_u32");
    run<2, 1>("ds_add_u64");      run<2, 0>("ds_add_u64");
    run<3, 1>("racy f32 rmw");    run<3, 0>("racy f32 rmw");
    run<5, 1>("ds_add_f64");      run<5, 0>("ds_add_f64");
    run<6, 0>("ds_add_u64 8-lane same addr"); run<7, 0>("ds_add_f32 8-lane same addr"); run<8, 0>("ds_add_f64 8-lane same addr");
    return 0;
}
