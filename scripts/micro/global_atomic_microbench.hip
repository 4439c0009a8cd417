// Microbenchmark: memory-side atomic throughput by type on MI355X (random addresses inside a 25 MB region, all lanes
// active, one atomic per lane per iteration; also a 64 KB region = heavy same-line contention).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o global_atomic_microbench global_atomic_microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void __launch_bounds__(256) k(int iters, unsigned mask, void* buf)
{
    float* f = (float*)buf; unsigned* u = (unsigned*)buf; unsigned long long* q = (unsigned long long*)buf;
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned a = (s >> 7) & mask;
        if (MODE == 0) unsafeAtomicAdd(&f[a], 1.0f);
        else if (MODE == 1) atomicAdd(&u[a], 1u);
        else atomicAdd(&q[a >> 1], 1ull);
    }
}

template <int MODE>
static void run(const char* name, unsigned mask, void* buf)
{
    const int iters = 256, blocks = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(iters, mask, buf); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(iters, mask, buf);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters;
    printf("%-10s region %8.2f MB  %.3f ms  %.1f G atomics/s\n", name, (mask + 1) * 4.0 / 1e6, ms, ops / ms / 1e6);
}

int main()
{
    void* buf; hipMalloc(&buf, 64 << 20); hipMemset(buf, 0, 64 << 20);
    for (unsigned mask : {(1u << 23) - 1u, (1u << 14) - 1u}) {
        run<0>("f32", mask, buf); run<1>("u32", mask, buf); run<2>("u64", mask, buf);
    }
    return 0;
}
