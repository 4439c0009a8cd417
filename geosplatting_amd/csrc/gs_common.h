// gs_common.h -- shared device/host helpers of libgeosplat_hip (gfx950 only; wave64 hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/geosplat_hip.h"

#define GS_WAVE 64

void gs_set_error(const char* fmt, ...);

#define GS_CHECK_ARG(cond, msg)                                  \
    do {                                                         \
        if (!(cond)) { gs_set_error("%s: %s", __func__, msg); return GS_EINVAL; } \
    } while (0)

#define GS_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            gs_set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(_e));  \
            return GS_ELAUNCH;                                                          \
        }                                                                               \
    } while (0)

#define GS_CHECK_LAUNCH()  GS_CHECK_HIP(hipGetLastError())

static inline int gs_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// A size that is either known on the host (dev == NULL: n itself) or lives in device memory (capacity protocol: n is the
// caller's CAPACITY -- grids and buffers are sized by it -- and the kernels read the actual count, clamped to the capacity).
struct GsCount {
    long long n;
    const long long* dev;
};
__device__ __forceinline__ long long gs_count(const GsCount& c)
{
    if (c.dev == nullptr) return c.n;
    const long long v = *c.dev;
    return v < 0 ? 0 : (v < c.n ? v : c.n);
}

// Zero device memory with a KERNEL, not hipMemsetAsync, on every path that may be captured into a HIP graph.  Observed on ROCm 7.2
// (scripts/debug_graph_sync4.py): when the replay of a captured step was launched behind an eager kernel that had already finished,
// the memset NODE that clears the chained-scan state of project_fwd_kernel was no longer ordered in front of that kernel -- its blocks
// then spun on look-back flags that were wiped under them (5-20 s per replay, memory faults later).  Kernel nodes keep their order.
static __global__ void __launch_bounds__(256) gs_zero_kernel(uint32_t* __restrict__ p, size_t n_words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static __global__ void __launch_bounds__(256) gs_zero16_kernel(uint4* __restrict__ p, size_t n_quads)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
static inline hipError_t gs_zero_async(void* p, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return hipSuccess;
    if ((bytes & 3) != 0 || (((uintptr_t)p) & 3) != 0) return hipMemsetAsync(p, 0, bytes, s);     // (no such caller on a graph path)
    const bool wide = (bytes & 15) == 0 && (((uintptr_t)p) & 15) == 0;
    const size_t n = wide ? bytes / 16 : bytes / 4;
    const size_t want = (n + 255) / 256;
    const int blocks = (int)(want < 4096 ? want : 4096);
    if (wide) hipLaunchKernelGGL(gs_zero16_kernel, dim3(blocks), dim3(256), 0, s, (uint4*)p, n);
    else hipLaunchKernelGGL(gs_zero_kernel, dim3(blocks), dim3(256), 0, s, (uint32_t*)p, n);
    return hipGetLastError();
}

// ---- wave64 cross-lane helpers -------------------------------------------------------------------
__device__ __forceinline__ float gs_readlane(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ int gs_readlane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float gs_dpp_add(float v)
{
    int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false);
    return v + __builtin_bit_cast(float, o);
}

// Sum over the 64 lanes of a wave; the result is valid in lane 63 (and returned broadcast via readlane).
__device__ __forceinline__ float gs_wave_sum(float v)
{
    v = gs_dpp_add<0xB1, 0xf, 0xf>(v);    // quad_perm [1,0,3,2]
    v = gs_dpp_add<0x4E, 0xf, 0xf>(v);    // quad_perm [2,3,0,1]
    v = gs_dpp_add<0x141, 0xf, 0xf>(v);   // row_half_mirror
    v = gs_dpp_add<0x140, 0xf, 0xf>(v);   // row_mirror  -> every lane of a 16-lane row holds the row sum
    v = gs_dpp_add<0x142, 0xa, 0xf>(v);   // row_bcast:15 into rows 1 and 3
    v = gs_dpp_add<0x143, 0xc, 0xf>(v);   // row_bcast:31 into rows 2 and 3
    return gs_readlane(v, 63);
}

// Inclusive prefix sum over the 64 lanes of a wave: row_shr:1/2/4/8 inside the 16-lane rows, row_bcast:15 and :31 across them, each
// move fused into its add -- six VALU instructions.  (`__shfl_up` compiles to ds_bpermute_b32: six dependent trips through the LDS
// crossbar, ~10x the latency, on kernels that are a few microseconds of dependent steps.)  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned gs_dpp_add_u32(unsigned v)
{
    return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);      // (lanes without a source add 0)
}
__device__ __forceinline__ unsigned gs_wave_incl_scan(unsigned v)
{
    v = gs_dpp_add_u32<0x111, 0xf>(v);    // row_shr:1
    v = gs_dpp_add_u32<0x112, 0xf>(v);    // row_shr:2
    v = gs_dpp_add_u32<0x114, 0xf>(v);    // row_shr:4
    v = gs_dpp_add_u32<0x118, 0xf>(v);    // row_shr:8   -> inclusive scan inside every row
    v = gs_dpp_add_u32<0x142, 0xa>(v);    // row_bcast:15: rows 1 and 3 += the total of the row before
    v = gs_dpp_add_u32<0x143, 0xc>(v);    // row_bcast:31: rows 2 and 3 += the total of rows 0 + 1
    return v;
}
__device__ __forceinline__ int gs_wave_incl_scan(int v) { return (int)gs_wave_incl_scan((unsigned)v); }
// Sum over the wave (every lane active), returned in all lanes through an SGPR
__device__ __forceinline__ unsigned gs_wave_sum_u32(unsigned v)
{
    v = gs_dpp_add_u32<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v = gs_dpp_add_u32<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    v = gs_dpp_add_u32<0x141, 0xf>(v);    // row_half_mirror
    v = gs_dpp_add_u32<0x140, 0xf>(v);    // row_mirror
    v = gs_dpp_add_u32<0x142, 0xa>(v);    // row_bcast:15
    v = gs_dpp_add_u32<0x143, 0xc>(v);    // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// ... of non-negative 64-bit values below 2^62: three 21-bit limbs, whose 64-lane sums fit 32 bits
__device__ __forceinline__ unsigned long long gs_wave_sum_u64(unsigned long long v)
{
    const unsigned a = gs_wave_sum_u32((unsigned)(v & 0x1fffffull)), b = gs_wave_sum_u32((unsigned)((v >> 21) & 0x1fffffull)),
                   c = gs_wave_sum_u32((unsigned)(v >> 42));
    return (unsigned long long)a + ((unsigned long long)b << 21) + ((unsigned long long)c << 42);
}

// ---- butterfly reduce-scatter over a wave --------------------------------------------------------------
// Sums NV per-lane values over the 64 lanes in ~3*NV/2 + 6 VALU ops instead of 6*NV: at every stage a lane
// hands HALF of its values to its partner and keeps the other half, so the value count halves while the
// partial sums double (DPP row_mirror, row_half_mirror, quad_perm xor2, quad_perm xor1 -- each fused into
// the add), then two cross-row exchanges.  Afterwards lane l (every row holds the same totals) owns the total
// of value index  16*q + 8*(l&1) + 4*((l>>1)&1) + 2*((l>>2)&1) + ((l>>3)&1)  in out[q].
template <int CTRL>
__device__ __forceinline__ float gs_dpp_take(float give)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, give), CTRL, 0xf, 0xf, true));
}

template <int N, int CTRL>
__device__ __forceinline__ void gs_bfly_stage(const float* in, float* out, bool own_hi)
{
    // pairs (in[2z], in[2z+1]) -> out[z]; a lane with own_hi keeps the odd member and gives the even one
#pragma unroll
    for (int z = 0; z < (N + 1) / 2; ++z) {
        const float lo = in[2 * z];
        const float hi = (2 * z + 1 < N) ? in[2 * z + 1] : 0.0f;
        const float keep = own_hi ? hi : lo;
        const float give = own_hi ? lo : hi;
        out[z] = keep + gs_dpp_take<CTRL>(give);
    }
}

template <int NV>
struct GsBfly {
    static constexpr int N1 = (NV + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2;
    // value index owned by this lane in out[q]
    __device__ static __forceinline__ int owned_index(int lane, int q)
    {
        return 16 * q + 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
    }
    // rows only: every 16-lane DPP row ends with ITS partial totals (no cross-row exchange) -- the caller lets
    // each row that had contributing lanes commit its own partial sums
    __device__ static __forceinline__ void reduce_rows(const float (&v)[NV], float (&out)[N4], int lane)
    {
        float a[N1], b[N2], c[N3];
        gs_bfly_stage<NV, 0x140>(v, a, (lane & 8) != 0);
        gs_bfly_stage<N1, 0x141>(a, b, (lane & 4) != 0);
        gs_bfly_stage<N2, 0x4E>(b, c, (lane & 2) != 0);
        gs_bfly_stage<N3, 0xB1>(c, out, (lane & 1) != 0);
    }
    __device__ static __forceinline__ void reduce(const float (&v)[NV], float (&out)[N4], int lane)
    {
        float a[N1], b[N2], c[N3];
        gs_bfly_stage<NV, 0x140>(v, a, (lane & 8) != 0);     // row_mirror       i <-> 15-i
        gs_bfly_stage<N1, 0x141>(a, b, (lane & 4) != 0);     // row_half_mirror  i <-> 7-i (within 8)
        gs_bfly_stage<N2, 0x4E>(b, c, (lane & 2) != 0);      // quad_perm [2,3,0,1]
        gs_bfly_stage<N3, 0xB1>(c, out, (lane & 1) != 0);    // quad_perm [1,0,3,2]
#pragma unroll
        for (int q = 0; q < N4; ++q) {                       // rows: xor 16, xor 32 through the LDS crossbar
            out[q] += __shfl_xor(out[q], 16, 64);
            out[q] += __shfl_xor(out[q], 32, 64);
        }
    }
};

// exp(-sigma) of the compositor in the canonical operation order shared with the oracle (oracle/gs_oracle.c gso_exp_neg):
// y = -sigma*log2(e), n = rint(y), degree-6 polynomial for 2^(y-n) in explicit FMAs, ldexp.  10 VALU operations instead of
// v_exp_f32's 1 quarter-rate one -- and alpha, T, every skip / stop decision, the image and last_ids come out bit-identical
// to the CPU oracle (v_exp_f32 differs from any libm in the last bits, which moved thresholds and made the stored-state
// backward's T_final = 1 - alpha differ by an ulp of 1.0, i.e. by 1e-3 of a saturated pixel's transmittance).
__device__ __forceinline__ float gs_exp_neg(float sigma)
{
#pragma clang fp contract(off)
    const float y = sigma * -1.44269504f;
    const float yc = fminf(fmaxf(y, -126.0f), 126.0f);
    const float n = __builtin_rintf(yc);
    const float f = yc - n;
    float p = 0x1.41a6fep-13f;
    p = __builtin_fmaf(p, f, 0x1.5f44f0p-10f);
    p = __builtin_fmaf(p, f, 0x1.3b2dfep-7f);
    p = __builtin_fmaf(p, f, 0x1.c6aed6p-5f);
    p = __builtin_fmaf(p, f, 0x1.ebfbdap-3f);
    p = __builtin_fmaf(p, f, 0x1.62e430p-1f);
    p = __builtin_fmaf(p, f, 1.0f);
    const float r = __builtin_ldexpf(p, (int)n);
    return (y >= -125.0f) ? r : 0.0f;
}

// extent of {alpha >= 1/255} for a Gaussian, conservatively inflated; returns false if it can never reach
__device__ __forceinline__ bool alpha_extent(float ca, float cb, float cc, float o, float& hx, float& hy)
{
    const float tau = __logf(255.0f * o);
    const float det = ca * cc - cb * cb;
    if (!(tau > -0.002f)) return false;              // o*255 < ~1: never visible (also rejects NaN)
    if (!(det > 0.0f)) { hx = hy = 1e30f; return true; }
    const float k = 2.0f * (tau + 0.002f) / det;
    hx = sqrtf(k * cc) * 1.0005f + 0.02f;
    hy = sqrtf(k * ca) * 1.0005f + 0.02f;
    return true;
}

// the 64-byte per-visible record the compositor's stream build gathers (one cache line per intersection):
//   {mx, my, 0.5a, b | 0.5c, opacity, hx, hy | c0, c1, c2, - | pad}      (colours only for D <= 3)
__device__ __forceinline__ void gs_write_vis_record(float4* __restrict__ rec, float mx, float my, float ca, float cb, float cc, float op,
                                                    float c0, float c1, float c2)
{
    float hx = -1.0f, hy = -1.0f;
    if (!alpha_extent(ca, cb, cc, op, hx, hy)) { hx = -1.0f; hy = -1.0f; }
    rec[0] = make_float4(mx, my, 0.5f * ca, cb);
    rec[1] = make_float4(0.5f * cc, op, hx, hy);
    rec[2] = make_float4(c0, c1, c2, 0.0f);
    rec[3] = make_float4(0.f, 0.f, 0.f, 0.f);             // full-line write (a partial line costs a read-modify-write)
}

__device__ __forceinline__ int gs_lane_id() { return (int)(threadIdx.x & 63); }

// fp32 atomic add that lowers to the hardware global_atomic_add_f32 (no CAS loop), agent (device) scope:
// coherent across the 8 XCDs, i.e. executed at the memory side of the fabric (a 32-byte write-through per op).
__device__ __forceinline__ void gs_atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }

// XCD-LOCAL fp32 atomic add (workgroup scope: no sc1 -> resolved in this XCD's L2, the line stays cached).
// ONLY valid on memory that no other XCD touches during the launch (per-XCD private accumulators).
__device__ __forceinline__ void gs_atomic_add_xcd(float* p, float v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// id of the XCD (accelerator complex die) this wave runs on, 0..7 on MI355X: HW_REG_XCC_ID (hwreg 20), bits [3:0]
__device__ __forceinline__ int gs_xcc_id()
{
    return (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf);
}
