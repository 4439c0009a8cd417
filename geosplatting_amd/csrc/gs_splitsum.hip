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

__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

__device__ __forceinline__ float ndfGGX(float alphaSqr, float cosTheta)
{
    const float c = fminf(fmaxf(cosTheta, 0.0f), 1.0f);
    const float d = (c * alphaSqr - c) * c + 1.0f;
    return alphaSqr / (d * d * 3.14159265358979323846f);
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
    GS_CHECK_ARG(n >= 0 && R >= 1, "bad sizes");
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
    GS_CHECK_ARG(R >= 1, "bad R");
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
    for (int s = 0; s < 6; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(bounds + (size_t)t * 24 + s * 4);
        const int xmin = (int)b.x, xmax = (int)b.y, ymin = (int)b.z, ymax = (int)b.w;
        if (xmin > xmax) continue;
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
                if (ldv >= cutoff) {
                    float Hv[3] = { L[0] + VNR[0], L[1] + VNR[1], L[2] + VNR[2] };
                    const float hl = sqrtf(dot3(Hv, Hv));
                    if (hl > 0.0f) { Hv[0] /= hl; Hv[1] /= hl; Hv[2] /= hl; } else { Hv[0] = Hv[1] = Hv[2] = 0.0f; }
                    const float wiDotN = fmaxf(ldv, 0.0f);
                    const float VNRDotH = fmaxf(dot3(VNR, Hv), 0.0f);
                    const float area = BWD ? own_area : o4.w;
                    const float w = wiDotN * ndfGGX(alphaSqr, VNRDotH) * area / 4.0f;
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

// ---------------------------------------------------------------------------------------------------
// Cached pair weights.  w(o,i) depends on (R, roughness, cutoff) only -- never on the cubemap -- and the cubemap
// is re-filtered every training step.  With 288 GB of HBM the weights are worth keeping: for every texel t the
// 8x8 patches of its face AABBs are stored as 64 contiguous floats in the traversal order of specular_kernel
// (0 outside the lobe / outside the AABB).  Applying the filter then is a pure stream: one coalesced 256-byte
// weight read per patch plus L2-resident texel gathers -- HBM-bound instead of division/sqrt-bound -- and the
// numbers are bit-identical to the direct kernel because the same kernel code fills the table.
// Two tables per level: forward (t = output texel) and backward (t = input texel, see specular_kernel<BWD>).
__global__ void __launch_bounds__(256)
specular_patch_count_kernel(int R, const float* __restrict__ bounds, int32_t* __restrict__ counts)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 6 * R * R) return;
    int n = 0;
    for (int s = 0; s < 6; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(bounds + (size_t)t * 24 + s * 4);
        const int xmin = (int)b.x, xmax = (int)b.y, ymin = (int)b.z, ymax = (int)b.w;
        if (xmin > xmax) continue;
        n += ((xmax - xmin) / 8 + 1) * ((ymax - ymin) / 8 + 1);
    }
    counts[t] = n;
}

extern "C" int gs_specular_patch_count(int R, const float* bounds, int32_t* counts, void* stream)
{
    GS_CHECK_ARG(R >= 1, "bad R");
    hipLaunchKernelGGL(specular_patch_count_kernel, dim3(gs_cdiv(6 * R * R, 256)), dim3(256), 0, (hipStream_t)stream, R,
                       bounds, counts);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// MODE 0: fill the weight table (and wsum in the forward orientation); MODE 1: apply a filled table.
template <bool BWD, int MODE>
__global__ void __launch_bounds__(256)
specular_table_kernel(int R, const float* __restrict__ src, const float* __restrict__ bounds,
                      const float4* __restrict__ table, const int64_t* __restrict__ patch_offsets,
                      float roughness, float cutoff, float* __restrict__ weights, float* __restrict__ wsum_out,
                      float* __restrict__ dst, int dst_stride, int accumulate, int32_t* __restrict__ patch_desc)
{
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= 6 * R * R) return;
    const int lx = lane & 7, ly = lane >> 3;
    float own[3] = { 0.f, 0.f, 0.f }, own_area = 0.0f, alphaSqr = 0.0f;
    if (MODE == 0) {
        const float4 own4 = table[t];
        own[0] = own4.x; own[1] = own4.y; own[2] = own4.z; own_area = own4.w;
        const float alpha = roughness * roughness;
        alphaSqr = alpha * alpha;
    }
    float* wp = weights + (size_t)patch_offsets[t] * 64 + lane;
    int32_t* dp = (MODE == 0 && patch_desc) ? patch_desc + patch_offsets[t] : nullptr;
    float wsum = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    for (int s = 0; s < 6; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(bounds + (size_t)t * 24 + s * 4);
        const int xmin = (int)b.x, xmax = (int)b.y, ymin = (int)b.z, ymax = (int)b.w;
        if (xmin > xmax) continue;
        for (int by = ymin; by <= ymax; by += 8)
            for (int bx = xmin; bx <= xmax; bx += 8, wp += 64) {
                const int x = bx + lx, y = by + ly;
                const bool in_box = (x <= xmax && y <= ymax);
                const size_t ti = ((size_t)s * R + y) * R + x;
                if (MODE == 0) {
                    float w = 0.0f;
                    if (in_box) {
                        const float4 o4 = table[ti];
                        const float other[3] = { o4.x, o4.y, o4.z };
                        const float* L = BWD ? own : other;
                        const float* VNR = BWD ? other : own;
                        const float ldv = dot3(L, VNR);
                        if (ldv >= cutoff) {
                            float Hv[3] = { L[0] + VNR[0], L[1] + VNR[1], L[2] + VNR[2] };
                            const float hl = sqrtf(dot3(Hv, Hv));
                            if (hl > 0.0f) { Hv[0] /= hl; Hv[1] /= hl; Hv[2] /= hl; } else { Hv[0] = Hv[1] = Hv[2] = 0.0f; }
                            const float wiDotN = fmaxf(ldv, 0.0f);
                            const float VNRDotH = fmaxf(dot3(VNR, Hv), 0.0f);
                            const float area = BWD ? own_area : o4.w;
                            w = wiDotN * ndfGGX(alphaSqr, VNRDotH) * area / 4.0f;
                        }
                    }
                    *wp = w;
                    wsum += w;
                    if (dp) { if (lane == 0) *dp = (s << 24) | (by << 12) | bx; ++dp; }
                } else {
                    const float w = *wp;
                    if (w != 0.0f) { c0 += src[ti * 3] * w; c1 += src[ti * 3 + 1] * w; c2 += src[ti * 3 + 2] * w; }
                }
            }
    }
    if (MODE == 0) {
        if (wsum_out) { wsum = gs_wave_sum(wsum); if (lane == 0) wsum_out[t] = wsum; }
    } else {
        c0 = gs_wave_sum(c0); c1 = gs_wave_sum(c1); c2 = gs_wave_sum(c2);
        if (lane == 0) {
            float* p = dst + (size_t)t * dst_stride;
            if (accumulate) { p[0] += c0; p[1] += c1; p[2] += c2; } else { p[0] = c0; p[1] = c1; p[2] = c2; }
        }
    }
}

extern "C" int gs_specular_weights_build(int R, const float* bounds, const float* dir_table, const int64_t* patch_offsets,
                                         float roughness, float costheta_cutoff, int backward, float* weights,
                                         float* wsum, int32_t* patch_desc, void* stream)
{
    GS_CHECK_ARG(R >= 1 && R < 4096 && dir_table && patch_offsets && weights, "bad arguments");
    const dim3 grid(gs_cdiv(6 * R * R, 4)), block(256);
    if (backward)
        hipLaunchKernelGGL((specular_table_kernel<true, 0>), grid, block, 0, (hipStream_t)stream, R, nullptr, bounds,
                           (const float4*)dir_table, patch_offsets, roughness, costheta_cutoff, weights, wsum, nullptr, 0, 0, patch_desc);
    else
        hipLaunchKernelGGL((specular_table_kernel<false, 0>), grid, block, 0, (hipStream_t)stream, R, nullptr, bounds,
                           (const float4*)dir_table, patch_offsets, roughness, costheta_cutoff, weights, wsum, nullptr, 0, 0, patch_desc);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// Streaming application of a weight table: the patches of texel t are a flat list (descriptor = face|by|bx),
// so four patches' weights and texels are requested before any is consumed (memory-level parallelism; the direct
// kernel's nested AABB loops serialise one patch's latency after the other).
// One load per lane and patch: three strided dword gathers bounded the first version (texture-addresser cycles, not HBM).
// SRC4 = float4-padded source [6,R,R,4], one 16-byte load; !SRC4 = the packed [6,R,R,3] map itself, one 12-byte load
// (dwordx3, 4-byte aligned) -- the default: a patch row of 8 texels is 96 instead of 128 bytes, i.e. fewer cache lines
// per tap instruction (1.67 vs 1.75 ms per direction), and the padded copy per level and direction disappears.
#ifndef GS_APPLY_XCD
#define GS_APPLY_XCD 1
#endif
#ifndef GS_APPLY_NT
#define GS_APPLY_NT 1
#endif
#if GS_APPLY_NT
#define GS_STREAM_LOAD(p) __builtin_nontemporal_load(p)
#else
#define GS_STREAM_LOAD(p) (*(p))
#endif
#ifndef GS_APPLY_UNROLL
#define GS_APPLY_UNROLL 4          // patches in flight per texel and wave (x GS_APPLY_TPW texels)
#endif
#ifndef GS_APPLY_WAVES
#define GS_APPLY_WAVES 4            // texels (waves) per workgroup
#endif
#ifndef GS_APPLY_LDS
#define GS_APPLY_LDS 0              // dynamic LDS request (occupancy cap experiment)
#endif
#ifndef GS_APPLY_TPW
#define GS_APPLY_TPW 2              // texels per wave, processed INTERLEAVED (their latency chains overlap)
#endif
// TPW texels per wave, interleaved: the offset loads, the weight batches and the tap batches of all TPW texels are
// issued together, so one wave carries TPW independent latency chains (offsets -> weights (HBM) -> taps -> sum); the taps
// are branch-free (zero-weight taps read texel 0 and are masked: a branch around a load makes the compiler wait for each
// tap before it issues the next -- the first version ran its 12 taps strictly one after the other).
// Measured (scripts/apply_experiment.py, ms per direction over the 6 levels): 12 serial taps 2.04; branch-free U=6 1.88;
// TPW=2 x U=4 1.76 (default); TPW=2 x 6 1.88; TPW=4 x 3 1.82.  What bounds it (scripts/run_pmc_apply.sh): the weight
// stream ALONE runs at 5.9-6.6 TB/s in this structure (a plain dword-per-lane stream: 7 TB/s, scripts/micro), the taps
// hit L1 (1.1 L2 requests per patch) but every tap instruction is 8+ cache-line accesses in the CU's in-order
// vector-memory pipeline behind the HBM-latency weight loads (latency-FIFO / pending stalls 30 % of the time, waves
// cannot issue 1/3 of their cycles): the two costs add instead of overlapping.  Dead ends: 16-byte loads throughout (one
// lane = 4 adjacent taps, 1 instead of 2 memory instructions per patch: 2.09-2.20), source regions staged through LDS
// per 16-texel workgroup so that only the weight stream uses the memory pipeline (2.63: the three-barrier prologue and
// the 58 staged texels per output texel cost more than the 211 taps they replace), workgroups of 8 / 16 waves (no L1
// gain), taps addressed independently of the weights (more lanes fetch: slower at 256^2).
template <bool SRC4>
__global__ void __launch_bounds__(64 * GS_APPLY_WAVES)
specular_apply_kernel(int R, const float* __restrict__ src, const int64_t* __restrict__ patch_offsets, int64_t total_patches,
                      const int32_t* __restrict__ patch_desc, const float* __restrict__ weights,
                      float* __restrict__ dst, int dst_stride, int accumulate, int t_begin, int t_end)
{
    constexpr int TPW = GS_APPLY_TPW, U = GS_APPLY_UNROLL;
    const int lane = threadIdx.x & 63;
#if GS_APPLY_XCD
    const int per = gridDim.x / 8;
    const int grp = (blockIdx.x % 8) * per + blockIdx.x / 8;
#else
    const int grp = blockIdx.x;
#endif
    // texels [t_begin, t_end) of the level: the whole level on one GPU, this rank's share when the prefilter is sharded
    const int t0 = t_begin + __builtin_amdgcn_readfirstlane((grp * GS_APPLY_WAVES + (int)(threadIdx.x >> 6)) * TPW);
    const int n = t_end;
    const int n_all = 6 * R * R;
    if (t0 >= n) return;
    const int lx = lane & 7, ly = lane >> 3;
    int64_t pb[TPW], pe[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = min(t0 + j, n - 1);
        pb[j] = patch_offsets[t];
        pe[j] = (t + 1 < n_all) ? patch_offsets[t + 1] : total_patches;
        if (t0 + j >= n) pe[j] = pb[j];
    }
    int64_t longest = 0;
#pragma unroll
    for (int j = 0; j < TPW; ++j) longest = max(longest, pe[j] - pb[j]);
    float c[TPW][3];
#pragma unroll
    for (int j = 0; j < TPW; ++j) c[j][0] = c[j][1] = c[j][2] = 0.0f;
    for (int64_t o = 0; o < longest; o += U) {
        float w[TPW][U]; unsigned ti[TPW][U];
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int64_t p = pb[j] + o + k;
                const bool on = p < pe[j];
                const int d = on ? patch_desc[p] : 0;
                w[j][k] = on ? GS_STREAM_LOAD(weights + (size_t)p * 64 + lane) : 0.0f;
                const int sf = d >> 24, by = (d >> 12) & 0xfff, bx = d & 0xfff;
                ti[j][k] = (unsigned)(((sf * R + (by + ly)) * R + (bx + lx)) * (SRC4 ? 4 : 3));
            }
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            float v[U][3];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const bool nz = w[j][k] != 0.0f;                     // zero-weight taps (outside the lobe / the face) read
                const unsigned a = nz ? ti[j][k] : 0u;               // texel 0 and are masked: no branch around the load
                if (SRC4) {
                    const float4 q = *reinterpret_cast<const float4*>(src + a);
                    v[k][0] = nz ? q.x : 0.0f; v[k][1] = nz ? q.y : 0.0f; v[k][2] = nz ? q.z : 0.0f;
                } else {
                    struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };     // one 12-byte load (dwordx3), 4-byte aligned
                    const F3 q = *reinterpret_cast<const F3*>(src + a);
                    v[k][0] = nz ? q.x : 0.0f; v[k][1] = nz ? q.y : 0.0f; v[k][2] = nz ? q.z : 0.0f;
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k) { c[j][0] += v[k][0] * w[j][k]; c[j][1] += v[k][1] * w[j][k]; c[j][2] += v[k][2] * w[j][k]; }
        }
    }
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const float s0 = gs_wave_sum(c[j][0]), s1 = gs_wave_sum(c[j][1]), s2 = gs_wave_sum(c[j][2]);
        if (lane == 0 && t0 + j < n) {
            float* q = dst + (size_t)(t0 + j) * dst_stride;
            if (accumulate) { q[0] += s0; q[1] += s1; q[2] += s2; } else { q[0] = s0; q[1] = s1; q[2] = s2; }
        }
    }
}

extern "C" int gs_specular_apply_range(int R, const float* src, int src_stride, const int64_t* patch_offsets,
                                       int64_t total_patches, const int32_t* patch_desc, const float* weights, float* dst,
                                       int dst_stride, int accumulate, int t_begin, int t_end, void* stream)
{
    GS_CHECK_ARG(R >= 1 && src && patch_offsets && patch_desc && weights && dst && dst_stride >= 3, "bad arguments");
    GS_CHECK_ARG(src_stride == 3 || src_stride == 4, "src_stride must be 3 or 4");
    GS_CHECK_ARG(t_begin >= 0 && t_begin <= t_end && t_end <= 6 * R * R, "texel range outside [0, 6 R^2]");
    if (t_begin == t_end) return GS_OK;
    const int groups = (gs_cdiv(t_end - t_begin, GS_APPLY_WAVES * GS_APPLY_TPW) + 7) / 8 * 8;   // multiple of 8: one contiguous share per XCD
    if (src_stride == 4)
        hipLaunchKernelGGL(specular_apply_kernel<true>, dim3(groups), dim3(64 * GS_APPLY_WAVES), GS_APPLY_LDS, (hipStream_t)stream, R, src,
                           patch_offsets, total_patches, patch_desc, weights, dst, dst_stride, accumulate, t_begin, t_end);
    else
        hipLaunchKernelGGL(specular_apply_kernel<false>, dim3(groups), dim3(64 * GS_APPLY_WAVES), GS_APPLY_LDS, (hipStream_t)stream, R, src,
                           patch_offsets, total_patches, patch_desc, weights, dst, dst_stride, accumulate, t_begin, t_end);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_specular_apply(int R, const float* src, int src_stride, const int64_t* patch_offsets,
                                 int64_t total_patches, const int32_t* patch_desc, const float* weights, float* dst,
                                 int dst_stride, int accumulate, void* stream)
{
    return gs_specular_apply_range(R, src, src_stride, patch_offsets, total_patches, patch_desc, weights, dst, dst_stride,
                                   accumulate, 0, 6 * R * R, stream);
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
