// gs_tone.h -- S4 tone mapping, per channel (rfstudio/model/geosplat.py:474-480: _tone_mapping_naive / _tone_mapping_aces; 'none' is the
// identity).  Shared by the stand-alone tone-map kernels (gs_shade.hip) and by the compositor kernels that apply it in their
// epilogue / prologue (gs_raster.hip: gs_raster_composite_tone*, gs_raster_bwd_tone*).  Contraction is OFF inside the functions so that
// every includer computes the same bits whatever its file-level setting is.
#pragma once
#include "gs_common.h"

__device__ __forceinline__ float tone_fwd(int mode, float rgb)
{
#pragma clang fp contract(off)
    if (mode == GS_TONE_NAIVE) {
        const float x = 1.0f - rgb, bx = 100.0f * x;
        const float sp = bx > 20.0f ? x : log1pf(expf(bx)) / 100.0f;
        return 1.0f - sp;
    }
    if (mode == GS_TONE_ACES) return (rgb * (2.51f * rgb + 0.03f)) / (rgb * (2.43f * rgb + 0.59f) + 0.14f);
    return rgb;
}
__device__ __forceinline__ float tone_grad(int mode, float rgb)
{
#pragma clang fp contract(off)
    if (mode == GS_TONE_NAIVE) {
        const float bx = 100.0f * (1.0f - rgb);
        return bx > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-bx));
    }
    if (mode == GS_TONE_ACES) {
        const float num = rgb * (2.51f * rgb + 0.03f), den = rgb * (2.43f * rgb + 0.59f) + 0.14f;
        return ((5.02f * rgb + 0.03f) * den - num * (4.86f * rgb + 0.59f)) / (den * den);
    }
    return 1.0f;
}
