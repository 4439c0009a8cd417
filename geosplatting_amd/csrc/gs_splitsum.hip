// gs_splitsum.hip -- S5: split-sum environment-map prefilter (once per training step, feeds S3).
// HIP equivalents of the reference's in-repo CUDA plugin `rfstudio_render_utils`:
//   DiffuseCubemapFwd/BwdKernel   rfstudio/graphics/_mesh/_splitsum/c_src/cubemap.cu:110-168
//   SpecularBoundsKernel          .../cubemap.cu:181-244
//   SpecularCubemapFwd/BwdKernel  .../cubemap.cu:246-350
//   _CubeMapMip fwd / bwd         rfstudio/graphics/_mesh/_texture.py:199-226
// Launch shapes are re-derived for wave64: one thread per output texel in 64x4 blocks over a flat texel
// index (the reference uses 8x8 blocks over (x,y,face)); the diffuse backward is formulated as a gather
// over input texels (no atomics, deterministic); the specular backward is a gather too (lobe membership is symmetric),
// so the whole prefilter backward is atomic-free and bit-reproducible.
#include "gs_common.h"

// Contraction OFF for the whole file: texel / LUT-cell / lobe-membership selection are discontinuous in the
// coordinates, so the coordinates are computed in the same one-rounding-per-operation order as the CPU oracle
// (these kernels are memory-bound; the lost FMAs cost nothing measurable).
#pragma clang fp contract(off)
#include "gs_cube.h"
#include "gs_splitsum_math.h"

// backward gathers look for the outputs that include a source texel inside the source's OWN lobe box grown by this margin
// (membership `ldv >= cutoff` is symmetric; the boxes, built from culled 16x16 tiles, are so only up to rare one-texel differences)
#define GS_SPECULAR_BWD_MARGIN 2

__device__ __forceinline__ float pixel_area(int x, int y, int N)
{
    if (N > 1) {
        const int H = N / 2;
        x = abs(x - H);
        y = abs(y - H);
        const float dx = atanf((float)(x + 1) / (float)H) - atanf((float)x / (float)H);
        const float dy = atanf((float)(y + 1) / (float)H) - atanf((float)y / (float)H);
        return dx * dy;
    }
    return 1.0f;
}

__device__ __forceinline__ void cube_to_dir(int x, int y, int side, int N, float* d)
{
    const float fx = 2.0f * (((float)x + 0.5f) / (float)N) - 1.0f;
    const float fy = 2.0f * (((float)y + 0.5f) / (float)N) - 1.0f;
    face_point(side, fx, fy, d);
    const float l = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (l > 0.0f) { d[0] /= l; d[1] /= l; d[2] /= l; } else { d[0] = d[1] = d[2] = 0.0f; }
}

// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
mip_fwd_kernel(int R, int C, const float* __restrict__ in, float* __restrict__ out)
{
    const int H = R / 2;
    const int64_t total = (int64_t)6 * H * H * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t t = i / C;
        const int x = (int)(t % H), y = (int)((t / H) % H), s = (int)(t / ((int64_t)H * H));
        const float* p = in + (((size_t)s * R + 2 * y) * R + 2 * x) * C + c;
        out[i] = (((p[0] + p[C]) + p[(size_t)R * C]) + p[(size_t)R * C + C]) * 0.25f;
    }
}

extern "C" int gs_cubemap_mip_fwd(int R, int C, const float* in, float* out, void* stream)
{
    GS_CHECK_ARG(R >= 2 && (R % 2) == 0 && C >= 1, "bad R/C");
    const int64_t total = (int64_t)6 * (R / 2) * (R / 2) * C;
    hipLaunchKernelGGL(mip_fwd_kernel, dim3((int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0,
                       (hipStream_t)stream, R, C, in, out);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// The whole mip chain R -> R/2 -> ... -> R/2^n in ONE launch (round 6; as n launches of mip_fwd_kernel the chain sat at the head of every
// step as five dependent 4-20 us kernels with 40-80 us between them while another queue was busy: profiles/r05_step_boundary.txt).
// A workgroup owns a 2^n x 2^n patch of one face: level 1 from global memory, every further level from the previous one in LDS
// (ping-pong), each level written out as it is formed.  Same four addends in the same order as mip_fwd_kernel: bit-identical.
#define GS_MIP_CHAIN_MAX 5
struct MipChainArgs { int R, n; float* out[GS_MIP_CHAIN_MAX]; };

__global__ void __launch_bounds__(256)
mip_chain_fwd_kernel(const float* __restrict__ in, const MipChainArgs a)
{
    __shared__ float buf[2][256 * 3];
    const int R = a.R, P = 1 << a.n, ppr = R / P;
    const int patch = blockIdx.x;
    const int face = patch / (ppr * ppr), py = (patch / ppr) % ppr, px = patch % ppr;
    const int t = threadIdx.x;
    int H = P >> 1;                                                   // patch edge at the level being formed
    int Rl = R >> 1;                                                  // face edge at that level
    if (t < H * H) {
        const int lx = t % H, ly = t / H;
        const float* p = in + (((size_t)face * R + (size_t)py * P + 2 * ly) * R + (size_t)px * P + 2 * lx) * 3;
        float* o = a.out[0] + (((size_t)face * Rl + (size_t)py * H + ly) * Rl + (size_t)px * H + lx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (((p[c] + p[3 + c]) + p[(size_t)R * 3 + c]) + p[(size_t)R * 3 + 3 + c]) * 0.25f;
            o[c] = v;
            buf[0][t * 3 + c] = v;
        }
    }
    for (int k = 1; k < a.n; ++k) {
        __syncthreads();
        const int Hp = H;                                             // edge of the level in LDS
        H >>= 1; Rl >>= 1;
        const float* src = buf[(k - 1) & 1];
        float* dstl = buf[k & 1];
        if (t < H * H) {
            const int lx = t % H, ly = t / H;
            const float* p = src + ((2 * ly) * Hp + 2 * lx) * 3;
            float* o = a.out[k] + (((size_t)face * Rl + (size_t)py * H + ly) * Rl + (size_t)px * H + lx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = (((p[c] + p[3 + c]) + p[Hp * 3 + c]) + p[Hp * 3 + 3 + c]) * 0.25f;
                o[c] = v;
                dstl[t * 3 + c] = v;
            }
        }
    }
}

extern "C" int gs_cubemap_mip_chain_fwd(int R, int n_levels, const float* in, float* const* outs, void* stream)
{
    GS_CHECK_ARG(n_levels >= 1 && n_levels <= GS_MIP_CHAIN_MAX && R >= (1 << n_levels) && (R % (1 << n_levels)) == 0 && in && outs,
                 "1..5 levels, R a multiple of 2^n_levels");
    MipChainArgs a;
    a.R = R; a.n = n_levels;
    for (int k = 0; k < GS_MIP_CHAIN_MAX; ++k) a.out[k] = k < n_levels ? outs[k] : nullptr;
    for (int k = 0; k < n_levels; ++k) GS_CHECK_ARG(a.out[k] != nullptr, "null output level");
    const int ppr = R >> n_levels;
    hipLaunchKernelGGL(mip_chain_fwd_kernel, dim3(6 * ppr * ppr), dim3(256), 0, (hipStream_t)stream, in, a);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

__global__ void __launch_bounds__(256)
cube_sample_kernel(int64_t n, const float* __restrict__ tex, int R, const float* __restrict__ dirs, float scale,
                   float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d[3] = { dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2] };
    float o[3]; CubeFp fp;
    cube_fetch<false>(tex, R, d, o, nullptr, fp);
    out[3 * i] = o[0] * scale; out[3 * i + 1] = o[1] * scale; out[3 * i + 2] = o[2] * scale;
}

extern "C" int gs_cube_sample_linear(int64_t n, const float* tex, int R, const float* dirs, float scale, float* out,
                                     void* stream)
{
    GS_CHECK_ARG(n >= 0 && R >= 1 && R <= GS_CUBE_EDGE_TABLE_MAX_R, "bad sizes (faces up to 4096^2: gs_cube.h)");
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(cube_sample_kernel, dim3(gs_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, tex, R, dirs,
                       scale, out);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// _CubeMapMip.backward: v_in[s,y,x] (+)= bilinear-cube(0.25 * v_out) at the direction of fine texel (x,y,s)
__global__ void __launch_bounds__(256)
mip_bwd_kernel(int R /*coarse*/, const float* __restrict__ v_out, float* __restrict__ v_in, int accumulate)
{
    const int F = 2 * R;
    const int64_t total = (int64_t)6 * F * F;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % F), y = (int)((i / F) % F), s = (int)(i / ((int64_t)F * F));
    // torch.linspace(-1 + 1/res, 1 - 1/res, res)[k]
    const float step = (2.0f - 2.0f / (float)F) / (float)(F - 1);
    const float gx = (-1.0f + 1.0f / (float)F) + step * (float)x;
    const float gy = (-1.0f + 1.0f / (float)F) + step * (float)y;
    float d[3];
    face_point(s, gx, gy, d);
    const float l = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] /= l; d[1] /= l; d[2] /= l;
    float o[3]; CubeFp fp;
    cube_fetch<false>(v_out, R, d, o, nullptr, fp);
    float* p = v_in + 3 * i;
    if (accumulate) { p[0] += 0.25f * o[0]; p[1] += 0.25f * o[1]; p[2] += 0.25f * o[2]; }
    else { p[0] = 0.25f * o[0]; p[1] = 0.25f * o[1]; p[2] = 0.25f * o[2]; }
}

extern "C" int gs_cubemap_mip_bwd(int R, const float* v_out, float* v_in, int accumulate, void* stream)
{
    GS_CHECK_ARG(R >= 1 && R <= GS_CUBE_EDGE_TABLE_MAX_R, "bad R (faces up to 4096^2: gs_cube.h)");
    const int64_t total = (int64_t)6 * 4 * R * R;
    hipLaunchKernelGGL(mip_bwd_kernel, dim3(gs_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, R, v_out, v_in,
                       accumulate);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// diffuse: out[o] = sum_i cubemap[i] * clamp(N_o . L_i, 0, 0.999) * area_i / 3.141592
// One 256-thread block per output texel; the 6*R*R inputs are strided over the threads and block-reduced
// (the reference runs one thread per output over all inputs: 1 536 threads only at R=16 -- 6 blocks on a
// 256-CU chip).  BWD is the same gather with the roles of input/output swapped (no atomics).
template <bool BWD>
__global__ void __launch_bounds__(256)
diffuse_kernel(int R, const float* __restrict__ src, float* __restrict__ dst, int accumulate)
{
    const int n = 6 * R * R;
    const int o = blockIdx.x;
    const int ox = o % R, oy = (o / R) % R, os = o / (R * R);
    float A[3]; cube_to_dir(ox, oy, os, R, A);
    const float pa_o = pixel_area(ox, oy, R);
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int x = i % R, y = (i / R) % R, s = i / (R * R);
        float B[3]; cube_to_dir(x, y, s, R, B);
        const float costheta = fminf(fmaxf(dot3(A, B), 0.0f), 0.999f);
        // fwd: weight uses the INPUT texel's area (i); bwd (gather over outputs i for input o): area of o
        const float w = costheta * (BWD ? pa_o : pixel_area(x, y, R)) / 3.141592f;
        const float* t = src + (size_t)i * 3;
        c0 += t[0] * w; c1 += t[1] * w; c2 += t[2] * w;
    }
    c0 = gs_wave_sum(c0); c1 = gs_wave_sum(c1); c2 = gs_wave_sum(c2);
    __shared__ float s_part[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_part[wave][0] = c0; s_part[wave][1] = c1; s_part[wave][2] = c2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        for (int w = 0; w < 4; ++w) { r0 += s_part[w][0]; r1 += s_part[w][1]; r2 += s_part[w][2]; }
        float* p = dst + (size_t)o * 3;
        if (accumulate) { p[0] += r0; p[1] += r1; p[2] += r2; } else { p[0] = r0; p[1] = r1; p[2] = r2; }
    }
}

extern "C" int gs_diffuse_cubemap_fwd(int R, const float* cubemap, float* out, void* stream)
{
    GS_CHECK_ARG(R >= 1 && R <= 64, "diffuse prefilter expects the 16^2 level (R <= 64)");
    hipLaunchKernelGGL(diffuse_kernel<false>, dim3(6 * R * R), dim3(256), 0, (hipStream_t)stream, R,
                       cubemap, out, 0);
    GS_CHECK_LAUNCH();
    return GS_OK;
}
extern "C" int gs_diffuse_cubemap_bwd(int R, const float* v_out, float* v_cubemap, int accumulate, void* stream)
{
    GS_CHECK_ARG(R >= 1 && R <= 64, "diffuse prefilter expects the 16^2 level (R <= 64)");
    hipLaunchKernelGGL(diffuse_kernel<true>, dim3(6 * R * R), dim3(256), 0, (hipStream_t)stream, R,
                       v_out, v_cubemap, accumulate);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
specular_bounds_kernel(int R, float cutoff, float* __restrict__ bounds)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= 6 * R * R) return;
    const int px = o % R, py = (o / R) % R, pz = o / (R * R);
    float VNR[3]; cube_to_dir(px, py, pz, R, VNR);
    const int TILE = 16;
    const int nt = (R + TILE - 1) / TILE;
    for (int s = 0; s < 6; ++s) {
        int min_x = R - 1, max_x = 0, min_y = R - 1, max_y = 0;
        for (int tx = 0; tx < nt; ++tx)
            for (int ty = 0; ty < nt; ++ty) {
                const int tsx = tx * TILE, tsy = ty * TILE;
                const int tex = min((tx + 1) * TILE, R), tey = min((ty + 1) * TILE, R);
                float L0[3], L1[3], L2[3], L3[3];
                cube_to_dir(tsx, tsy, s, R, L0); cube_to_dir(tex, tsy, s, R, L1);
                cube_to_dir(tsx, tey, s, R, L2); cube_to_dir(tex, tey, s, R, L3);
                float maxdp = 0.0f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float mn = fminf(fminf(L0[k], L1[k]), fminf(L2[k], L3[k]));
                    const float mx = fmaxf(fmaxf(L0[k], L1[k]), fmaxf(L2[k], L3[k]));
                    maxdp += fmaxf(mn * VNR[k], mx * VNR[k]);
                }
                if (maxdp >= cutoff) {
                    for (int y = tsy; y < tey; ++y)
                        for (int x = tsx; x < tex; ++x) {
                            float L[3]; cube_to_dir(x, y, s, R, L);
                            if (dot3(L, VNR) >= cutoff) {
                                min_x = min(min_x, x); max_x = max(max_x, x);
                                min_y = min(min_y, y); max_y = max(max_y, y);
                            }
                        }
                }
            }
        float* b = bounds + (size_t)o * 24 + s * 4;
        b[0] = (float)min_x; b[1] = (float)max_x; b[2] = (float)min_y; b[3] = (float)max_y;
    }
}

extern "C" int gs_specular_bounds(int R, float costheta_cutoff, float* bounds, void* stream)
{
    GS_CHECK_ARG(R >= 1, "bad R");
    hipLaunchKernelGGL(specular_bounds_kernel, dim3(gs_cdiv(6 * R * R, 256)), dim3(256), 0, (hipStream_t)stream, R,
                       costheta_cutoff, bounds);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// The same boxes, two orders of magnitude faster (223 ms -> ~2 ms for the six levels of a 512^2 pyramid).  specular_bounds_kernel
// above is shaped like the reference's SpecularBoundsKernel (cubemap.cu:181-244): every texel re-derives the four normalised
// corner directions of every 16x16 tile of every face (4 x 6144 normalisations per texel at R = 512) before it tests the tile.
//  * bounds_tile_aabb_kernel computes the corner-direction box {mn[3], mx[3]} of every tile ONCE -- the same fminf / fmaxf of the
//    same four cube_to_dir values -- and of every GROUP of 4x4 tiles (component-wise min / max of its tiles' boxes);
//  * specular_bounds_fast_kernel evaluates the reference's tile test `sum_k max(mn_k V_k, mx_k V_k) >= cutoff` from those boxes
//    in the reference's operation order, so every tile decision is bit-identical; a group whose own test fails is skipped as a
//    whole: its box contains its tiles' boxes, each product and each sum is monotone under rounding, so a failing group test
//    implies that every tile test in it fails;
//  * the texel test of a surviving tile reads the cached direction table (bit-identical to cube_to_dir, dir_table_kernel).
__global__ void __launch_bounds__(256)
bounds_tile_aabb_kernel(int R, int nt, int ng, float* __restrict__ tile_box /*[6][nt][nt][6]*/, float* __restrict__ group_box /*[6][ng][ng][6]*/)
{
    const int TILE = 16;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_tiles = 6 * nt * nt, n_groups = 6 * ng * ng;
    if (i < n_tiles) {
        const int tx = i % nt, ty = (i / nt) % nt, s = i / (nt * nt);
        const int tsx = tx * TILE, tsy = ty * TILE;
        const int tex = min((tx + 1) * TILE, R), tey = min((ty + 1) * TILE, R);
        float L0[3], L1[3], L2[3], L3[3];
        cube_to_dir(tsx, tsy, s, R, L0); cube_to_dir(tex, tsy, s, R, L1);
        cube_to_dir(tsx, tey, s, R, L2); cube_to_dir(tex, tey, s, R, L3);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            tile_box[(size_t)i * 6 + k] = fminf(fminf(L0[k], L1[k]), fminf(L2[k], L3[k]));
            tile_box[(size_t)i * 6 + 3 + k] = fmaxf(fmaxf(L0[k], L1[k]), fmaxf(L2[k], L3[k]));
        }
    }
    // groups of 4x4 tiles: recomputed from the corner directions (no dependency on the tile entries written above)
    if (i < n_groups) {
        const int gx = i % ng, gy = (i / ng) % ng, s = i / (ng * ng);
        float mn[3] = { 3.0f, 3.0f, 3.0f }, mx[3] = { -3.0f, -3.0f, -3.0f };
        for (int ty = gy * 4; ty < min(gy * 4 + 4, nt); ++ty)
            for (int tx = gx * 4; tx < min(gx * 4 + 4, nt); ++tx) {
                const int tsx = tx * TILE, tsy = ty * TILE;
                const int tex = min((tx + 1) * TILE, R), tey = min((ty + 1) * TILE, R);
                float L0[3], L1[3], L2[3], L3[3];
                cube_to_dir(tsx, tsy, s, R, L0); cube_to_dir(tex, tsy, s, R, L1);
                cube_to_dir(tsx, tey, s, R, L2); cube_to_dir(tex, tey, s, R, L3);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    mn[k] = fminf(mn[k], fminf(fminf(L0[k], L1[k]), fminf(L2[k], L3[k])));
                    mx[k] = fmaxf(mx[k], fmaxf(fmaxf(L0[k], L1[k]), fmaxf(L2[k], L3[k])));
                }
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) { group_box[(size_t)i * 6 + k] = mn[k]; group_box[(size_t)i * 6 + 3 + k] = mx[k]; }
    }
}

__device__ __forceinline__ float box_maxdp(const float* __restrict__ box, const float* V)
{
    float maxdp = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) maxdp += fmaxf(box[k] * V[k], box[3 + k] * V[k]);
    return maxdp;
}

// One WAVE per output texel (VNR is wave-uniform): lanes test 64 groups, then the <= 16 tiles of a surviving group, then 64 texels
// of a surviving tile per step -- uniform control flow, coalesced table reads -- and keep per-lane min / max that meet in one
// cross-lane reduction per face.  (One THREAD per texel, the reference's shape, diverged: 32 ms for the six levels.)
__device__ __forceinline__ int bounds_wave_min(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int bounds_wave_max(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ void __launch_bounds__(256)
specular_bounds_fast_kernel(int R, int nt, int ng, float cutoff, const float4* __restrict__ table, const float* __restrict__ tile_box,
                            const float* __restrict__ group_box, float* __restrict__ bounds)
{
    const int lane = threadIdx.x & 63;
    const int o = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (o >= 6 * R * R) return;
    const float4 own = table[o];
    const float VNR[3] = { own.x, own.y, own.z };
    const int TILE = 16;
    for (int s = 0; s < 6; ++s) {
        int min_x = R - 1, max_x = 0, min_y = R - 1, max_y = 0;
        for (int g0 = 0; g0 < ng * ng; g0 += 64) {
            const int g = g0 + lane;
            const bool gpass = g < ng * ng && box_maxdp(group_box + (size_t)(s * ng * ng + g) * 6, VNR) >= cutoff;
            unsigned long long gmask = __ballot(gpass);
            while (gmask != 0ull) {
                const int gi = g0 + __builtin_ctzll(gmask);
                gmask &= gmask - 1ull;
                const int gx = gi % ng, gy = gi / ng;
                const int tx = gx * 4 + (lane & 3), ty = gy * 4 + ((lane >> 2) & 3);
                const bool tpass = lane < 16 && tx < nt && ty < nt &&
                                   box_maxdp(tile_box + (size_t)((s * nt + ty) * nt + tx) * 6, VNR) >= cutoff;
                unsigned long long tmask = __ballot(tpass);
                while (tmask != 0ull) {
                    const int tl = __builtin_ctzll(tmask);
                    tmask &= tmask - 1ull;
                    const int tsx = (gx * 4 + (tl & 3)) * TILE, tsy = (gy * 4 + (tl >> 2)) * TILE;
                    const int tex = min(tsx + TILE, R), tey = min(tsy + TILE, R);
                    const int x = tsx + (lane & 15);
                    for (int y = tsy + (lane >> 4); y < tey; y += 4) {
                        if (x < tex) {
                            const float4 q = table[((size_t)s * R + y) * R + x];
                            const float L[3] = { q.x, q.y, q.z };
                            if (dot3(L, VNR) >= cutoff) {
                                min_x = min(min_x, x); max_x = max(max_x, x);
                                min_y = min(min_y, y); max_y = max(max_y, y);
                            }
                        }
                    }
                }
            }
        }
        min_x = bounds_wave_min(min_x); max_x = bounds_wave_max(max_x);
        min_y = bounds_wave_min(min_y); max_y = bounds_wave_max(max_y);
        if (lane == 0)
            *reinterpret_cast<float4*>(bounds + (size_t)o * 24 + s * 4) = make_float4((float)min_x, (float)max_x, (float)min_y, (float)max_y);
    }
}

extern "C" size_t gs_specular_bounds_ws_bytes(int R)
{
    if (R < 1) return 0;
    const size_t nt = (size_t)(R + 15) / 16, ng = (nt + 3) / 4;
    return (6 * nt * nt + 6 * ng * ng) * 6 * sizeof(float);
}

extern "C" int gs_specular_bounds_fast(int R, float costheta_cutoff, const float* dir_table, float* bounds, void* ws, size_t ws_bytes,
                                       void* stream)
{
    GS_CHECK_ARG(R >= 1 && dir_table && bounds && ws, "bad arguments");
    GS_CHECK_ARG(ws_bytes >= gs_specular_bounds_ws_bytes(R), "workspace too small (gs_specular_bounds_ws_bytes)");
    const int nt = (R + 15) / 16, ng = (nt + 3) / 4;
    float* tile_box = (float*)ws;
    float* group_box = tile_box + (size_t)6 * nt * nt * 6;
    hipLaunchKernelGGL(bounds_tile_aabb_kernel, dim3(gs_cdiv(6 * nt * nt, 256)), dim3(256), 0, (hipStream_t)stream, R, nt, ng, tile_box, group_box);
    GS_CHECK_LAUNCH();
    hipLaunchKernelGGL(specular_bounds_fast_kernel, dim3(gs_cdiv(6 * R * R, 4)), dim3(256), 0, (hipStream_t)stream, R, nt, ng,
                       costheta_cutoff, (const float4*)dir_table, tile_box, group_box, bounds);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// Per-texel table {dir.xyz, pixel_area}: depends on R only, cached by the host across steps.  It removes the
// normalisation (3 correctly-rounded divisions + sqrt) and the four atanf of pixel_area from every (output,
// input) pair of the lobe loops; values are bit-identical to calling cube_to_dir / pixel_area in place.
__global__ void __launch_bounds__(256)
dir_table_kernel(int R, float4* __restrict__ table)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 6 * R * R) return;
    const int px = t % R, py = (t / R) % R, pz = t / (R * R);
    float d[3]; cube_to_dir(px, py, pz, R, d);
    table[t] = make_float4(d[0], d[1], d[2], pixel_area(px, py, R));
}

extern "C" int gs_cube_dir_table(int R, float* table, void* stream)
{
    GS_CHECK_ARG(R >= 1 && table != nullptr, "bad R/table");
    hipLaunchKernelGGL(dir_table_kernel, dim3(gs_cdiv(6 * R * R, 256)), dim3(256), 0, (hipStream_t)stream, R, (float4*)table);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// One WAVE per texel t (4 texels per 256-thread block); the 64 lanes tile each face's AABB in 8x8 patches.
// FWD: t is the OUTPUT texel (VNR = dir_t) and gathers the input texels of its lobe.
// BWD: t is the INPUT texel (L = dir_t) and gathers the OUTPUT texels whose lobe contains it -- lobe
// membership dot(L,VNR) >= cutoff is symmetric and the per-texel AABB table is a function of the direction
// only, so bounds[t] serves both roles.  The pair weight w(o,i) is evaluated with exactly the forward's
// operands (VNR = output direction, pixel_area of the input texel), which makes the backward the exact
// adjoint of the forward WITHOUT atomics (the reference scatters with atomicAdd, cubemap.cu:300-350).
// (The reference runs one THREAD per output texel: 1 536 threads at the 16^2 level that each walk the whole
// cube map.  One wave per texel keeps >= 1 536 waves in flight at every level.)
template <bool BWD>
__global__ void __launch_bounds__(256)
specular_kernel(int R, const float* __restrict__ src /*cubemap (fwd) | v_out rgb (bwd)*/,
                const float* __restrict__ bounds, const float4* __restrict__ table, float roughness, float cutoff,
                float* __restrict__ dst, int accumulate)
{
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= 6 * R * R) return;
    const float4 own4 = table[t];
    const float own[3] = { own4.x, own4.y, own4.z };
    const float own_area = own4.w;
    const float alpha = roughness * roughness;
    const float alphaSqr = alpha * alpha;
    const int lx = lane & 7, ly = lane >> 3;
    float wsum = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    const int own_s = t / (R * R), own_y = (t / R) % R, own_x = t % R;
    for (int s = 0; s < 6; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(bounds + (size_t)t * 24 + s * 4);
        int xmin = (int)b.x, xmax = (int)b.y, ymin = (int)b.z, ymax = (int)b.w;
        if (BWD && R < 64) {
            // small levels: one or four 16x16 tiles per face, where the reference's tile culling (whose boxes these are) is far
            // from symmetric -- every texel of every face is a candidate output, the pair test below decides
            xmin = ymin = 0; xmax = ymax = R - 1;
        } else {
            if (xmin > xmax) continue;
            if (BWD) {  // candidate outputs: this texel's own box plus a margin
                xmin = max(xmin - GS_SPECULAR_BWD_MARGIN, 0); xmax = min(xmax + GS_SPECULAR_BWD_MARGIN, R - 1);
                ymin = max(ymin - GS_SPECULAR_BWD_MARGIN, 0); ymax = min(ymax + GS_SPECULAR_BWD_MARGIN, R - 1);
            }
        }
        for (int by = ymin; by <= ymax; by += 8)
            for (int bx = xmin; bx <= xmax; bx += 8) {
                const int x = bx + lx, y = by + ly;
                if (x > xmax || y > ymax) continue;
                const size_t ti = ((size_t)s * R + y) * R + x;
                const float4 o4 = table[ti];
                const float other[3] = { o4.x, o4.y, o4.z };
                const float* L = BWD ? own : other;
                const float* VNR = BWD ? other : own;
                const float ldv = dot3(L, VNR);
                bool in = ldv >= cutoff;
                if (BWD && in) {
                    // exact adjoint of the forward: output `ti` gathers THIS texel only if it lies inside ti's box on this face
                    // (the reference's backward is the forward loop with a scatter, cubemap.cu:300-350)
                    const float4 ob = *reinterpret_cast<const float4*>(bounds + ti * 24 + own_s * 4);
                    in = own_x >= (int)ob.x && own_x <= (int)ob.y && own_y >= (int)ob.z && own_y <= (int)ob.w;
                }
                if (in) {
                    const float area = BWD ? own_area : o4.w;
                    const float w = specular_pair_g(L, VNR, ldv, alphaSqr) * area / 4.0f;
                    c0 += src[ti * 3] * w; c1 += src[ti * 3 + 1] * w; c2 += src[ti * 3 + 2] * w;
                    wsum += w;
                }
            }
    }
    c0 = gs_wave_sum(c0); c1 = gs_wave_sum(c1); c2 = gs_wave_sum(c2);
    if (!BWD) wsum = gs_wave_sum(wsum);
    if (lane == 0) {
        if (BWD) {
            float* p = dst + (size_t)t * 3;
            if (accumulate) { p[0] += c0; p[1] += c1; p[2] += c2; } else { p[0] = c0; p[1] = c1; p[2] = c2; }
        } else {
            *reinterpret_cast<float4*>(dst + (size_t)t * 4) = make_float4(c0, c1, c2, wsum);
        }
    }
}

extern "C" int gs_specular_cubemap_fwd(int R, const float* cubemap, const float* bounds, const float* dir_table,
                                       float roughness, float costheta_cutoff, float* out, void* stream)
{
    GS_CHECK_ARG(R >= 1 && dir_table != nullptr, "bad R / dir_table");
    hipLaunchKernelGGL(specular_kernel<false>, dim3(gs_cdiv(6 * R * R, 4)), dim3(256), 0, (hipStream_t)stream, R,
                       cubemap, bounds, (const float4*)dir_table, roughness, costheta_cutoff, out, 0);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_specular_cubemap_bwd(int R, const float* bounds, const float* dir_table, const float* v_out_rgb,
                                       float roughness, float costheta_cutoff, float* v_cubemap, int accumulate,
                                       void* stream)
{
    GS_CHECK_ARG(R >= 1 && dir_table != nullptr, "bad R / dir_table");
    hipLaunchKernelGGL(specular_kernel<true>, dim3(gs_cdiv(6 * R * R, 4)), dim3(256), 0, (hipStream_t)stream, R,
                       v_out_rgb, bounds, (const float4*)dir_table, roughness, costheta_cutoff, v_cubemap, accumulate);
    GS_CHECK_LAUNCH();
    return GS_OK;
}
