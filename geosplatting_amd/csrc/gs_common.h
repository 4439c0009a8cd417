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

__device__ __forceinline__ int gs_lane_id() { return (int)(threadIdx.x & 63); }

// fp32 atomic add that lowers to the hardware global_atomic_add_f32 (no CAS loop)
__device__ __forceinline__ void gs_atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
