// gs_splitsum_math.h -- the pair arithmetic of the split-sum specular prefilter in ONE place (gs_splitsum.hip: direct and
// per-texel table kernels; gs_splitsum_tiles.hip: tiled tables).  Operation order == rfstudio/graphics/_mesh/_splitsum/c_src/
// cubemap.cu:246-298 as restated by oracle/gs_oracle_splitsum.c; every includer compiles it under
// `#pragma clang fp contract(off)` (one rounding per operation, IEEE division and square root), because lobe membership is a
// threshold on `ldv` and, at the 512^2 level (alpha^2 = 4e-5), ONE ulp of VNRDotH moves a weight by 0.3-0.6 %.
#pragma once
#include "gs_common.h"

__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

__device__ __forceinline__ float ndfGGX(float alphaSqr, float cosTheta)
{
    const float c = fminf(fmaxf(cosTheta, 0.0f), 1.0f);
    const float d = (c * alphaSqr - c) * c + 1.0f;
    return alphaSqr / (d * d * 3.14159265358979323846f);
}

// g(o, i) = max(L.VNR, 0) * D_GGX(alpha^2, max(VNR.H, 0)),  H = normalize(L + VNR): the part of the pair weight
//   w(o, i) = g(o, i) * pixel_area(i) / 4      (cubemap.cu:286-289: `wiDotN * ndfGGX(..) * pixel_area(..) / 4.0f`, left to right)
// that does not depend on the source texel's area.  L = direction of the source texel i, VNR = direction of the output texel o,
// ldv = dot3(L, VNR) (the caller has it from the lobe test).  Every operation in it is invariant under a sign flip of one
// world axis applied to BOTH directions (products of two flipped components, sums of flipped addends): g of a mirrored pair
// has the same bits -- gs_splitsum_tiles.hip stores it once per orbit of the cube's three reflections.
__device__ __forceinline__ float specular_pair_g(const float* L, const float* VNR, float ldv, float alphaSqr)
{
    float Hv[3] = { L[0] + VNR[0], L[1] + VNR[1], L[2] + VNR[2] };
    const float hl = sqrtf(dot3(Hv, Hv));
    if (hl > 0.0f) { Hv[0] /= hl; Hv[1] /= hl; Hv[2] /= hl; } else { Hv[0] = Hv[1] = Hv[2] = 0.0f; }
    const float wiDotN = fmaxf(ldv, 0.0f);
    const float VNRDotH = fmaxf(dot3(VNR, Hv), 0.0f);
    return wiDotN * ndfGGX(alphaSqr, VNRDotH);
}
