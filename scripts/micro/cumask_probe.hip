// Which XCDs / CUs does a stream created with hipExtStreamCreateWithCUMask use?   hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
// Prints, per mask under test, the histogram of workgroups over (XCC_ID, SE, CU) read from the hardware registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void probe(uint32_t* out)
{
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // spin a little so that the blocks spread over every CU the mask allows
    long long t0 = clock64();
    while (clock64() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
int main()
{
    const int nb = 4096;
    uint32_t* d; CHECK(hipMalloc(&d, nb * 8));
    std::vector<uint32_t> h(2 * nb);
    struct T { const char* name; uint32_t m[8]; } tests[] = {
        { "bits 0..31", { 0xffffffffu, 0, 0, 0, 0, 0, 0, 0 } },
        { "bits 0..63", { 0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0 } },
        { "every 8th bit (0, 8, 16, ...)", { 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u } },
        { "bits with (i % 8) < 2", { 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u, 0x03030303u } },
        { "all", { ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u } },
    };
    for (auto& t : tests) {
        hipStream_t s;
        CHECK(hipExtStreamCreateWithCUMask(&s, 8, t.m));
        CHECK(hipMemsetAsync(d, 0xff, nb * 8, s));
        hipLaunchKernelGGL(probe, dim3(nb), dim3(64), 0, s, d);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
        std::map<uint32_t, int> xcc; std::map<uint64_t, int> cu;
        for (int i = 0; i < nb; ++i) {
            const uint32_t x = h[2 * i] & 0xf, hw = h[2 * i + 1];
            const uint32_t cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            ++xcc[x]; ++cu[((uint64_t)x << 32) | (se << 8) | (sh << 4) | cu_id];
        }
        printf("%-34s: %zu distinct CUs; workgroups per XCC:", t.name, cu.size());
        for (auto& kv : xcc) printf(" %u:%d", kv.first, kv.second);
        printf("\n");
        CHECK(hipStreamDestroy(s));
    }
    return 0;
}
