// Microbenchmark: what HBM read rate does a weight-stream-shaped access reach on MI355X?
//   shape: every wave reads `patches` consecutive 256-byte rows (one wave = one "texel" of the prefilter apply kernel)
//   W=1: one dword per lane per load instruction (256 B per wave instruction, U loads in flight)
//   W=4: one dwordx4 per lane (1 KiB per wave instruction, U/4 loads in flight)
// build: hipcc --offload-arch=gfx950 -O3 -o stream_microbench stream_microbench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) k_dword(const float* __restrict__ w, int patches, long total_patches, float* out)
{
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long p0 = t * patches;
    if (p0 >= total_patches) return;
    float acc = 0.f;
    for (int p = 0; p < patches; p += U) {
        float v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const float* a = w + (p0 + p + k) * 64 + lane;
            v[k] = (p + k < patches) ? (NT ? __builtin_nontemporal_load(a) : *a) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) acc += v[k];
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int U4, bool NT>
__global__ void __launch_bounds__(256) k_dwordx4(const f4* __restrict__ w, int patches, long total_patches, float* out)
{
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long p0 = t * patches;
    if (p0 >= total_patches) return;
    float acc = 0.f;
    for (int p = 0; p < patches; p += 4 * U4) {          // 4 patches per wave instruction
        f4 v[U4];
#pragma unroll
        for (int k = 0; k < U4; ++k) {
            const f4* a = w + (p0 + p + 4 * k) * 16 + lane;
            v[k] = (p + 4 * k < patches) ? (NT ? __builtin_nontemporal_load(a) : *a) : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < U4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <typename F>
static void run(const char* name, long total_patches, int patches, F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s patches/wave=%3d  %.3f ms  %.2f TB/s\n", name, patches, ms, total_patches * 256.0 / ms / 1e9);
}

int main()
{
    const long total = 12L << 20;                        // 12 Mi patches = 3.2 GB
    float* w; float* out;
    hipMalloc(&w, (total + 4096) * 256); hipMalloc(&out, 4);
    hipMemset(w, 0, total * 256);
    for (int patches : {8, 12, 24, 96, 768}) {
        const long waves = total / patches;
        const int blocks = (int)((waves + 3) / 4);
        run("dword  U=12 nt", total, patches, [&] { k_dword<12, true><<<blocks, 256>>>(w, patches, total, out); });
        run("dword  U=12", total, patches, [&] { k_dword<12, false><<<blocks, 256>>>(w, patches, total, out); });
        run("dword  U=6  nt", total, patches, [&] { k_dword<6, true><<<blocks, 256>>>(w, patches, total, out); });
        run("dwordx4 U4=3 nt", total, patches, [&] { k_dwordx4<3, true><<<blocks, 256>>>((const f4*)w, patches, total, out); });
        run("dwordx4 U4=3", total, patches, [&] { k_dwordx4<3, false><<<blocks, 256>>>((const f4*)w, patches, total, out); });
        run("dwordx4 U4=6 nt", total, patches, [&] { k_dwordx4<6, true><<<blocks, 256>>>((const f4*)w, patches, total, out); });
    }
    return 0;
}
