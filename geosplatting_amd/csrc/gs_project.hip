// gs_project.hip -- per-Gaussian stages of the rasterizer half of the path (gfx950):
//   A1  fused projection + cull + anti-alias compensation, packed by a single-pass chained scan
//   A1' opacity*compensation and colour gather
//   A2  tiles-per-Gaussian, inclusive tile cumsum, key/value emission
//   A4  per-tile offsets
//   A7  projection backward + gather backward
// Semantics: gsplat 1.4 `rasterization(packed=True, rasterize_mode='antialiased')` as called by the
// reference at rfstudio/model/gsplat.py:334-355.  All of these stages are
// HBM-bound streaming kernels (44 B in / ~60 B out per Gaussian): one thread per Gaussian, coalesced
// row reads, packed outputs written through an in-launch decoupled look-back scan so that every input
// array is read exactly once.
//
// Floating point: the forward projection is compiled with contraction OFF and spelled in the canonical
// operation order documented in DESIGN.md so that radii / tile ranges / sort keys are bit-exact against
// the CPU oracle.  The backward may contract.
#include "gs_common.h"

#include <stdarg.h>
#include <string.h>

// ---------------------------------------------------------------------------------------------------
// error string (thread-local)
static thread_local char g_err[512] = "";
void gs_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* gs_last_error(void) { return g_err; }
extern "C" int gs_version(void) { return 100; }

// ---------------------------------------------------------------------------------------------------
struct GsCam {
    float R[9];
    float t[3];
    float fx, fy, cx, cy;
};

__device__ __forceinline__ GsCam load_cam(const float* __restrict__ viewmat, const float* __restrict__ K)
{
    GsCam c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) c.R[i * 3 + j] = viewmat[i * 4 + j];
        c.t[i] = viewmat[i * 4 + 3];
    }
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
    return c;
}

struct ProjFwd {
    bool valid;
    int radius;
    float m2x, m2y, depth, ca, cb, cc, comp;
};

#pragma clang fp contract(off)
__device__ __forceinline__ void quat_to_rotmat_exact(float qw, float qx, float qy, float qz, float* R)
{
    float n2 = ((qx * qx + qy * qy) + qz * qz) + qw * qw;
    float inv = 1.0f / sqrtf(n2);
    float x = qx * inv, y = qy * inv, z = qz * inv, w = qw * inv;
    float x2 = x * x, y2 = y * y, z2 = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.0f - 2.0f * (y2 + z2); R[1] = 2.0f * (xy - wz);        R[2] = 2.0f * (xz + wy);
    R[3] = 2.0f * (xy + wz);        R[4] = 1.0f - 2.0f * (x2 + z2); R[5] = 2.0f * (yz - wx);
    R[6] = 2.0f * (xz - wy);        R[7] = 2.0f * (yz + wx);        R[8] = 1.0f - 2.0f * (x2 + y2);
}

// Canonical-order forward projection of one Gaussian (bit-exact twin of project_one in the oracle).
__device__ __forceinline__ ProjFwd project_exact(const GsCam& c, const float* mean, const float* quat,
                                                 const float* scale, float Wf, float Hf, float eps2d,
                                                 float near_plane, float far_plane, float radius_clip)
{
    ProjFwd o;
    o.valid = false; o.radius = 0;
    o.m2x = o.m2y = o.depth = o.ca = o.cb = o.cc = o.comp = 0.0f;
    float mc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        mc[i] = ((c.R[i * 3 + 0] * mean[0] + c.R[i * 3 + 1] * mean[1]) + c.R[i * 3 + 2] * mean[2]) + c.t[i];
    if (mc[2] < near_plane || mc[2] > far_plane) return o;

    float Rq[9], M[9], cov[9], T1[9], Cc[9];
    quat_to_rotmat_exact(quat[0], quat[1], quat[2], quat[3], Rq);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * scale[j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            cov[i * 3 + j] = (M[i * 3 + 0] * M[j * 3 + 0] + M[i * 3 + 1] * M[j * 3 + 1]) + M[i * 3 + 2] * M[j * 3 + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            T1[i * 3 + j] = (c.R[i * 3 + 0] * cov[0 * 3 + j] + c.R[i * 3 + 1] * cov[1 * 3 + j]) + c.R[i * 3 + 2] * cov[2 * 3 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Cc[i * 3 + j] = (T1[i * 3 + 0] * c.R[j * 3 + 0] + T1[i * 3 + 1] * c.R[j * 3 + 1]) + T1[i * 3 + 2] * c.R[j * 3 + 2];

    float x = mc[0], y = mc[1], z = mc[2];
    float tan_fovx = 0.5f * Wf / c.fx;
    float tan_fovy = 0.5f * Hf / c.fy;
    float lim_x_pos = (Wf - c.cx) / c.fx + 0.3f * tan_fovx;
    float lim_x_neg = c.cx / c.fx + 0.3f * tan_fovx;
    float lim_y_pos = (Hf - c.cy) / c.fy + 0.3f * tan_fovy;
    float lim_y_neg = c.cy / c.fy + 0.3f * tan_fovy;
    float rz = 1.0f / z;
    float rz2 = rz * rz;
    float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, x * rz));
    float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, y * rz));
    float J00 = c.fx * rz, J02 = -c.fx * tx * rz2;
    float J11 = c.fy * rz, J12 = -c.fy * ty * rz2;
    float A00 = J00 * Cc[0] + J02 * Cc[6], A01 = J00 * Cc[1] + J02 * Cc[7], A02 = J00 * Cc[2] + J02 * Cc[8];
    float A10 = J11 * Cc[3] + J12 * Cc[6], A11 = J11 * Cc[4] + J12 * Cc[7], A12 = J11 * Cc[5] + J12 * Cc[8];
    float c00 = A00 * J00 + A02 * J02;
    float c01 = A01 * J11 + A02 * J12;
    float c10 = A10 * J00 + A12 * J02;
    float c11 = A11 * J11 + A12 * J12;
    float m2x = c.fx * x * rz + c.cx;
    float m2y = c.fy * y * rz + c.cy;

    float det_orig = c00 * c11 - c01 * c10;
    c00 += eps2d; c11 += eps2d;
    float det = c00 * c11 - c01 * c10;
    float comp = sqrtf(fmaxf(0.0f, det_orig / det));
    if (det <= 0.0f) return o;
    float inv_det = 1.0f / det;
    float ca = c11 * inv_det, cb = -c01 * inv_det, cc = c00 * inv_det;

    float b = 0.5f * (c00 + c11);
    float v1 = b + sqrtf(fmaxf(0.01f, b * b - det));
    float radius = ceilf(3.0f * sqrtf(v1));
    if (radius <= radius_clip) return o;
    if (m2x + radius <= 0.0f || m2x - radius >= Wf || m2y + radius <= 0.0f || m2y - radius >= Hf) return o;

    o.valid = true;
    o.radius = (int)radius;
    o.m2x = m2x; o.m2y = m2y; o.depth = z;
    o.ca = ca; o.cb = cb; o.cc = cc; o.comp = comp;
    return o;
}

__device__ __forceinline__ void tile_range_exact(float mx, float my, int radius, int tile_size, int tw, int th,
                                                 int& x0, int& y0, int& x1, int& y1)
{
    float ts = (float)tile_size;
    float tr = (float)radius / ts;
    float tx = mx / ts, ty = my / ts;
    float fx0 = floorf(tx - tr), fy0 = floorf(ty - tr);
    float fx1 = ceilf(tx + tr), fy1 = ceilf(ty + tr);
    x0 = fx0 < 0.0f ? 0 : (fx0 > (float)tw ? tw : (int)fx0);
    y0 = fy0 < 0.0f ? 0 : (fy0 > (float)th ? th : (int)fy0);
    x1 = fx1 < 0.0f ? 0 : (fx1 > (float)tw ? tw : (int)fx1);
    y1 = fy1 < 0.0f ? 0 : (fy1 > (float)th ? th : (int)fy1);
}
// (contraction stays off for the rest of the file: the backward mirrors the oracle's rounding as well;
//  every kernel here is HBM-bound, the lost FMAs are free)

// ---------------------------------------------------------------------------------------------------
// Chained-scan state (one per launch, zeroed by gs_zero_async -- a kernel, see gs_common.h -- before the launch):
//   word 0      : ticket counter (chunk ids are handed out in ARRIVAL order -> look-back cannot deadlock)
//   word 1      : error flag (spin timeout)
//   then 4 arrays of n_chunks u64, each word written exactly once: bit 63 = valid, bits 62..0 = value
//     agg_v, agg_i : this chunk's own (visible count, tile count)
//     pre_v, pre_i : inclusive prefix up to and including this chunk
// Every descriptor word is an aligned 8-byte granule written by one relaxed agent-scope atomic store and
// read with relaxed agent-scope atomic loads (L1-bypassing, the "data is the flag" hand-off): no fence is
// needed because nothing but the granule itself crosses workgroups.
#ifndef GS_PROJ_BLOCK
#define GS_PROJ_BLOCK 512         // chunk of the chained scan = workgroup (alone: 1024 is 10 % faster than 256; under the three-stream overlap of the engine the 1024-thread blocks wait for residency next to the LDS-heavy compositor blocks: 512 -> +3 % views/s); 256 -> 1024 quarters the same-address ticket
#endif                             // atomics and descriptor traffic of the look-back: 0.185 -> 0.135 ms (stage incl. glue)
#define GS_PROJ_WAVES (GS_PROJ_BLOCK / 64)
#define GS_VALID_BIT  (1ull << 63)
#define GS_SPIN_LIMIT (1 << 22)

typedef unsigned long long u64;

__device__ __forceinline__ void desc_store(u64* p, u64 v)
{
    __hip_atomic_store(p, v | GS_VALID_BIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 desc_load(u64* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ void __launch_bounds__(GS_PROJ_BLOCK)
project_fwd_kernel(int N, const float* __restrict__ means, const float* __restrict__ quats,
                   const float* __restrict__ scales, const float* __restrict__ opacities,
                   const float* __restrict__ colors, int D, const float* __restrict__ viewmat,
                   const float* __restrict__ K, int W, int H, int tile_size, int tile_w, int tile_h,
                   float eps2d, float near_plane, float far_plane, float radius_clip,
                   int32_t* __restrict__ gaussian_ids, int32_t* __restrict__ radii, float* __restrict__ means2d,
                   float* __restrict__ depths, float* __restrict__ conics, float* __restrict__ compensations,
                   float* __restrict__ opacities_packed, float* __restrict__ colors_packed,
                   int32_t* __restrict__ tiles_per_gauss, int64_t* __restrict__ cum_tiles,
                   int32_t* __restrict__ packed_index,
                   unsigned* __restrict__ ctrl, u64* __restrict__ desc, int n_chunks, int64_t* __restrict__ counts, float4* __restrict__ vis)
{
    __shared__ int s_chunk;
    __shared__ int s_wv[GS_PROJ_WAVES];
    __shared__ int s_wi[GS_PROJ_WAVES];
    __shared__ long long s_base[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_chunk = (int)atomicAdd(ctrl, 1u);
    __syncthreads();
    const int chunk = s_chunk;
    const int n = chunk * GS_PROJ_BLOCK + tid;

    u64* agg_v = desc;
    u64* agg_i = desc + n_chunks;
    u64* pre_v = desc + 2 * (size_t)n_chunks;
    u64* pre_i = desc + 3 * (size_t)n_chunks;

    GsCam cam = load_cam(viewmat, K);
    ProjFwd p;
    p.valid = false; p.radius = 0;
    int ntiles = 0;
    int tx0 = 0, ty0 = 0, tx1 = 0, ty1 = 0;
    if (n < N) {
        float mean[3] = { means[3 * (size_t)n], means[3 * (size_t)n + 1], means[3 * (size_t)n + 2] };
        float4 q = *reinterpret_cast<const float4*>(quats + 4 * (size_t)n);
        float quat[4] = { q.x, q.y, q.z, q.w };
        float scale[3] = { scales[3 * (size_t)n], scales[3 * (size_t)n + 1], scales[3 * (size_t)n + 2] };
        p = project_exact(cam, mean, quat, scale, (float)W, (float)H, eps2d, near_plane, far_plane, radius_clip);
        if (p.valid) {
            tile_range_exact(p.m2x, p.m2y, p.radius, tile_size, tile_w, tile_h, tx0, ty0, tx1, ty1);
            ntiles = (tx1 - tx0) * (ty1 - ty0);
        }
    }

    // ---- block-level scan of (valid, ntiles)
    const u64 bal = __ballot(p.valid);
    const int v_excl_wave = __popcll(bal & ((1ull << lane) - 1ull));
    int i_incl = ntiles;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(i_incl, off, 64);
        if (lane >= off) i_incl += t;
    }
    if (lane == 63) { s_wv[wave] = __popcll(bal); s_wi[wave] = i_incl; }
    __syncthreads();
    int v_before = 0, i_before = 0, aggV = 0, aggI = 0;
#pragma unroll
    for (int w = 0; w < GS_PROJ_WAVES; ++w) {
        if (w < wave) { v_before += s_wv[w]; i_before += s_wi[w]; }
        aggV += s_wv[w]; aggI += s_wi[w];
    }

    // ---- decoupled look-back across chunks (wave 0)
    if (wave == 0) {
        long long baseV = 0, baseI = 0;
        if (chunk > 0) {
            if (lane == 0) { desc_store(&agg_v[chunk], (u64)aggV); desc_store(&agg_i[chunk], (u64)aggI); }
            int pos = chunk - 1;
            for (;;) {
                const int idx = pos - lane;
                bool isP = idx < 0;           // virtual predecessors before chunk 0: prefix 0
                bool isA = false;
                u64 pv = 0, pi = 0, av = 0, ai = 0;
                if (idx >= 0) {
                    int spins = 0;
                    for (;;) {
                        pv = desc_load(&pre_v[idx]); pi = desc_load(&pre_i[idx]);
                        isP = (pv & pi & GS_VALID_BIT) != 0;
                        if (isP) break;
                        av = desc_load(&agg_v[idx]); ai = desc_load(&agg_i[idx]);
                        isA = (av & ai & GS_VALID_BIT) != 0;
                        if (isA) break;
                        if (++spins > GS_SPIN_LIMIT) { atomicExch(ctrl + 1, 1u); isA = true; av = ai = 0; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                const u64 pmask = __ballot(isP);
                const int first = pmask ? __builtin_ctzll(pmask) : 64;
                long long cv = 0, ci = 0;
                if (lane <= first) {
                    cv = (long long)((isP ? pv : av) & ~GS_VALID_BIT);
                    ci = (long long)((isP ? pi : ai) & ~GS_VALID_BIT);
                }
                baseV += wave_sum_i64(cv);
                baseI += wave_sum_i64(ci);
                if (pmask) break;
                pos -= 64;
            }
        }
        if (lane == 0) {
            desc_store(&pre_i[chunk], (u64)(baseI + aggI));
            desc_store(&pre_v[chunk], (u64)(baseV + aggV));
            s_base[0] = baseV; s_base[1] = baseI;
            if (chunk == n_chunks - 1) { counts[0] = baseV + aggV; counts[1] = baseI + aggI; }
        }
    }
    __syncthreads();
    const long long baseV = s_base[0], baseI = s_base[1];

    if (n < N) {
        const long long slot = baseV + v_before + v_excl_wave;
        if (packed_index) packed_index[n] = p.valid ? (int32_t)slot : -1;
        if (p.valid) {
            gaussian_ids[slot] = n;
            radii[slot] = p.radius;
            *reinterpret_cast<float2*>(means2d + 2 * slot) = make_float2(p.m2x, p.m2y);
            depths[slot] = p.depth;
            conics[3 * slot] = p.ca; conics[3 * slot + 1] = p.cb; conics[3 * slot + 2] = p.cc;
            compensations[slot] = p.comp;
            opacities_packed[slot] = opacities[n] * p.comp;
            tiles_per_gauss[slot] = ntiles;
            cum_tiles[slot] = baseI + i_before + i_incl;
            if (colors_packed) {
                for (int k = 0; k < D; ++k) colors_packed[slot * D + k] = colors[(size_t)n * D + k];
            }
            if (vis) {                                       // the compositor's per-visible record, straight from the registers
                float c0 = 0.f, c1 = 0.f, c2 = 0.f;
                if (colors && D <= 3) {
                    c0 = colors[(size_t)n * D];
                    if (D > 1) c1 = colors[(size_t)n * D + 1];
                    if (D > 2) c2 = colors[(size_t)n * D + 2];
                }
                gs_write_vis_record(vis + 4 * slot, p.m2x, p.m2y, p.ca, p.cb, p.cc, opacities[n] * p.comp, c0, c1, c2);
            }
        }
    }
}

extern "C" size_t gs_project_ws_bytes(int N)
{
    size_t n_chunks = (size_t)((N + GS_PROJ_BLOCK - 1) / GS_PROJ_BLOCK);
    if (n_chunks == 0) n_chunks = 1;
    return 16 + 4 * n_chunks * sizeof(u64);
}

extern "C" int gs_project_fwd(int N, const float* means, const float* quats, const float* scales,
                              const float* opacities, const float* colors, int D, const float* viewmat,
                              const float* K, int W, int H, int tile_size, float eps2d, float near_plane,
                              float far_plane, float radius_clip, int32_t* gaussian_ids, int32_t* radii,
                              float* means2d, float* depths, float* conics, float* compensations,
                              float* opacities_packed, float* colors_packed, int32_t* tiles_per_gauss,
                              int64_t* cum_tiles, int32_t* packed_index, void* ws, size_t ws_bytes,
                              int64_t* counts, void* stream)
{
    return gs_project_fwd_vis(N, means, quats, scales, opacities, colors, D, viewmat, K, W, H, tile_size, eps2d, near_plane,
                              far_plane, radius_clip, gaussian_ids, radii, means2d, depths, conics, compensations,
                              opacities_packed, colors_packed, tiles_per_gauss, cum_tiles, packed_index, nullptr, ws, ws_bytes,
                              counts, stream);
}

extern "C" int gs_project_fwd_vis(int N, const float* means, const float* quats, const float* scales,
                                  const float* opacities, const float* colors, int D, const float* viewmat,
                                  const float* K, int W, int H, int tile_size, float eps2d, float near_plane,
                                  float far_plane, float radius_clip, int32_t* gaussian_ids, int32_t* radii,
                                  float* means2d, float* depths, float* conics, float* compensations,
                                  float* opacities_packed, float* colors_packed, int32_t* tiles_per_gauss,
                                  int64_t* cum_tiles, int32_t* packed_index, float* vis_records, void* ws, size_t ws_bytes,
                                  int64_t* counts, void* stream)
{
    GS_CHECK_ARG(N >= 0 && W > 0 && H > 0 && tile_size > 0, "bad sizes");
    GS_CHECK_ARG(counts != nullptr && ws != nullptr, "counts/ws must not be NULL");
    GS_CHECK_ARG((colors == nullptr) == (colors_packed == nullptr), "colors and colors_packed go together");
    if (ws_bytes < gs_project_ws_bytes(N)) { gs_set_error("gs_project_fwd: workspace too small"); return GS_ENOSPC; }
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_HIP(gs_zero_async(ws, gs_project_ws_bytes(N), s));
    GS_CHECK_HIP(gs_zero_async(counts, 2 * sizeof(int64_t), s));
    if (N == 0) return GS_OK;
    const int n_chunks = (N + GS_PROJ_BLOCK - 1) / GS_PROJ_BLOCK;
    const int tile_w = (W + tile_size - 1) / tile_size, tile_h = (H + tile_size - 1) / tile_size;
    unsigned* ctrl = (unsigned*)ws;
    u64* desc = (u64*)((char*)ws + 16);
    hipLaunchKernelGGL(project_fwd_kernel, dim3(n_chunks), dim3(GS_PROJ_BLOCK), 0, s, N, means, quats, scales,
                       opacities, colors, D, viewmat, K, W, H, tile_size, tile_w, tile_h, eps2d, near_plane,
                       far_plane, radius_clip, gaussian_ids, radii, means2d, depths, conics, compensations,
                       opacities_packed, colors_packed, tiles_per_gauss, cum_tiles, packed_index, ctrl, desc,
                       n_chunks, counts, (float4*)vis_records);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// A2 emit: one thread per visible Gaussian.
__global__ void __launch_bounds__(256)
isect_emit_kernel(int V, const float* __restrict__ means2d, const int32_t* __restrict__ radii,
                  const float* __restrict__ depths, const int64_t* __restrict__ cum_tiles, int tile_size,
                  int tile_w, int tile_h, int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int r = radii[v];
    if (r <= 0) return;
    const float2 m = *reinterpret_cast<const float2*>(means2d + 2 * (size_t)v);
    int x0, y0, x1, y1;
    tile_range_exact(m.x, m.y, r, tile_size, tile_w, tile_h, x0, y0, x1, y1);
    int64_t cur = (v == 0) ? 0 : cum_tiles[v - 1];
    const int64_t depth_enc = (int64_t)(uint32_t)__float_as_int(depths[v]);
    for (int i = y0; i < y1; ++i)
        for (int j = x0; j < x1; ++j) {
            const int64_t tile_id = (int64_t)i * tile_w + j;
            isect_ids[cur] = (tile_id << 32) | depth_enc;
            flatten_ids[cur] = v;
            ++cur;
        }
}

extern "C" int gs_isect_emit(int V, const float* means2d, const int32_t* radii, const float* depths,
                             const int64_t* cum_tiles, int tile_size, int tile_w, int tile_h,
                             int64_t* isect_ids, int32_t* flatten_ids, void* stream)
{
    GS_CHECK_ARG(V >= 0 && tile_size > 0, "bad sizes");
    if (V == 0) return GS_OK;
    hipLaunchKernelGGL(isect_emit_kernel, dim3(gs_cdiv(V, 256)), dim3(256), 0, (hipStream_t)stream, V, means2d,
                       radii, depths, cum_tiles, tile_size, tile_w, tile_h, isect_ids, flatten_ids);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// A4 offsets: offsets[t] = first sorted position whose tile id >= t.
__global__ void __launch_bounds__(256)
isect_offsets_kernel(GsCount nc, const int64_t* __restrict__ ids, int n_tiles, int32_t* __restrict__ offsets)
{
    const int64_t n = gs_count(nc);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 && i == 0)                                  // (only reachable with a device-side count: the host path memsets)
        for (int t = 0; t < n_tiles; ++t) offsets[t] = 0;
    if (i >= n) return;
    const int cur = (int)(((uint64_t)ids[i]) >> 32);
    if (i == 0) {
        for (int t = 0; t <= cur && t < n_tiles; ++t) offsets[t] = 0;
    } else {
        const int prev = (int)(((uint64_t)ids[i - 1]) >> 32);
        for (int t = prev + 1; t <= cur && t < n_tiles; ++t) offsets[t] = (int32_t)i;
    }
    if (i == n - 1)
        for (int t = cur + 1; t < n_tiles; ++t) offsets[t] = (int32_t)n;
}

// the same from int32 tile ids (gs_isect_bin_tiles_cap)
__global__ void __launch_bounds__(256)
isect_offsets_tiles_kernel(GsCount nc, const int32_t* __restrict__ tiles, int n_tiles, int32_t* __restrict__ offsets)
{
    const int64_t n = gs_count(nc);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 && i == 0)
        for (int t = 0; t < n_tiles; ++t) offsets[t] = 0;
    if (i >= n) return;
    const int cur = tiles[i];
    if (i == 0) {
        for (int t = 0; t <= cur && t < n_tiles; ++t) offsets[t] = 0;
    } else {
        const int prev = tiles[i - 1];
        for (int t = prev + 1; t <= cur && t < n_tiles; ++t) offsets[t] = (int32_t)i;
    }
    if (i == n - 1)
        for (int t = cur + 1; t < n_tiles; ++t) offsets[t] = (int32_t)n;
}

extern "C" int gs_isect_offsets_tiles_cap(int64_t n_isects_cap, const int64_t* counts_dev, const int32_t* tile_ids_sorted, int n_tiles,
                                          int32_t* offsets, void* stream)
{
    GS_CHECK_ARG(n_isects_cap > 0 && n_tiles > 0 && counts_dev != nullptr && tile_ids_sorted != nullptr, "bad arguments");
    hipLaunchKernelGGL(isect_offsets_tiles_kernel, dim3(gs_cdiv(n_isects_cap, 256)), dim3(256), 0, (hipStream_t)stream,
                       GsCount{ n_isects_cap, (const long long*)counts_dev + 1 }, tile_ids_sorted, n_tiles, offsets);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_isect_offsets(int64_t n_isects, const int64_t* isect_ids_sorted, int n_tiles, int32_t* offsets,
                                void* stream)
{
    GS_CHECK_ARG(n_isects >= 0 && n_tiles > 0, "bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (n_isects == 0) {
        GS_CHECK_HIP(gs_zero_async(offsets, sizeof(int32_t) * (size_t)n_tiles, s));
        return GS_OK;
    }
    hipLaunchKernelGGL(isect_offsets_kernel, dim3(gs_cdiv(n_isects, 256)), dim3(256), 0, s, GsCount{ n_isects, nullptr },
                       isect_ids_sorted, n_tiles, offsets);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_isect_offsets_cap(int64_t n_isects_cap, const int64_t* counts_dev, const int64_t* isect_ids_sorted, int n_tiles,
                                    int32_t* offsets, void* stream)
{
    GS_CHECK_ARG(n_isects_cap > 0 && n_tiles > 0 && counts_dev != nullptr, "bad sizes");
    hipLaunchKernelGGL(isect_offsets_kernel, dim3(gs_cdiv(n_isects_cap, 256)), dim3(256), 0, (hipStream_t)stream,
                       GsCount{ n_isects_cap, (const long long*)counts_dev + 1 }, isect_ids_sorted, n_tiles, offsets);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// A7: projection backward + gather backward.  One thread per INPUT Gaussian so that the dense [N,*]
// gradients are written coalesced and exactly once (culled Gaussians get zeros, no memset pass); the
// packed slot comes from a binary search in the ascending gaussian_ids list (log2(V) L2-resident probes).
__device__ __forceinline__ int find_slot(const int32_t* __restrict__ gids, int V, int N, int n)
{
    // gids is ascending and holds V of the N indices, so the slot of n lies in [n - (N - V), n]: the search collapses to
    // zero probes when every Gaussian is visible and to log2(N - V + 1) dependent L2 probes otherwise (was log2 V = 21)
    int lo = max(0, n - (N - V)), hi = min(n + 1, V);
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (gids[mid] < n) lo = mid + 1; else hi = mid;
    }
    return (lo < V && gids[lo] == n) ? lo : -1;
}

__global__ void __launch_bounds__(256)
project_bwd_kernel(int N, GsCount vc, int D, const float* __restrict__ means, const float* __restrict__ quats,
                   const float* __restrict__ scales, const float* __restrict__ opacities,
                   const float* __restrict__ viewmat, const float* __restrict__ K, int W, int H, float eps2d,
                   const int32_t* __restrict__ gaussian_ids, const float* __restrict__ conics,
                   const float* __restrict__ compensations, const float* __restrict__ v_packed, int rec_stride,
                   const float* __restrict__ v_depths,
                   float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
                   float* __restrict__ v_opacities, float* __restrict__ v_colors, int accumulate)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int V = (int)gs_count(vc);
    const int v = find_slot(gaussian_ids, V, N, n);
    float g_mean[3] = { 0, 0, 0 }, g_quat[4] = { 0, 0, 0, 0 }, g_scale[3] = { 0, 0, 0 }, g_op = 0.0f;
    if (v >= 0) {
        const GsCam cam = load_cam(viewmat, K);
        const float mean[3] = { means[3 * (size_t)n], means[3 * (size_t)n + 1], means[3 * (size_t)n + 2] };
        const float4 q4 = *reinterpret_cast<const float4*>(quats + 4 * (size_t)n);
        const float scale[3] = { scales[3 * (size_t)n], scales[3 * (size_t)n + 1], scales[3 * (size_t)n + 2] };
        const float comp = compensations[v];
        const float* __restrict__ rec = v_packed + (size_t)v * rec_stride;   // {xy(2), conic(3), opacity, colors(D)}
        const float v_op = rec[5];
        g_op = v_op * comp;
        const float v_comp = v_op * opacities[n];

        // recompute forward intermediates
        float mc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            mc[i] = cam.R[i * 3 + 0] * mean[0] + cam.R[i * 3 + 1] * mean[1] + cam.R[i * 3 + 2] * mean[2] + cam.t[i];
        float Rq[9], M[9], cov[9], T1[9], Cc[9];
        quat_to_rotmat_exact(q4.x, q4.y, q4.z, q4.w, Rq);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * scale[j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                cov[i * 3 + j] = M[i * 3 + 0] * M[j * 3 + 0] + M[i * 3 + 1] * M[j * 3 + 1] + M[i * 3 + 2] * M[j * 3 + 2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                T1[i * 3 + j] = cam.R[i * 3 + 0] * cov[0 * 3 + j] + cam.R[i * 3 + 1] * cov[1 * 3 + j] + cam.R[i * 3 + 2] * cov[2 * 3 + j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                Cc[i * 3 + j] = T1[i * 3 + 0] * cam.R[j * 3 + 0] + T1[i * 3 + 1] * cam.R[j * 3 + 1] + T1[i * 3 + 2] * cam.R[j * 3 + 2];

        // conic = inverse(cov2d_blur): v_cov2d = -conic * v_conic_mat * conic
        const float ia = conics[3 * (size_t)v], ib = conics[3 * (size_t)v + 1], ic = conics[3 * (size_t)v + 2];
        const float ga = rec[2], gb = 0.5f * rec[3], gc = rec[4];
        const float p00 = ia * ga + ib * gb, p01 = ia * gb + ib * gc;
        const float p10 = ib * ga + ic * gb, p11 = ib * gb + ic * gc;
        float G[4] = { -(p00 * ia + p01 * ib), -(p00 * ib + p01 * ic), -(p10 * ia + p11 * ib), -(p10 * ib + p11 * ic) };
        {   // compensation vjp
            const float det_conic = ia * ic - ib * ib;
            const float v_sqr_comp = v_comp * 0.5f / (comp + 1e-6f);
            const float om = 1.0f - comp * comp;
            G[0] += v_sqr_comp * (om * ia - eps2d * det_conic);
            G[1] += v_sqr_comp * (om * ib);
            G[2] += v_sqr_comp * (om * ib);
            G[3] += v_sqr_comp * (om * ic - eps2d * det_conic);
        }
        // perspective projection vjp
        const float x = mc[0], y = mc[1], z = mc[2];
        const float Wf = (float)W, Hf = (float)H;
        const float tan_fovx = 0.5f * Wf / cam.fx, tan_fovy = 0.5f * Hf / cam.fy;
        const float lim_x_pos = (Wf - cam.cx) / cam.fx + 0.3f * tan_fovx;
        const float lim_x_neg = cam.cx / cam.fx + 0.3f * tan_fovx;
        const float lim_y_pos = (Hf - cam.cy) / cam.fy + 0.3f * tan_fovy;
        const float lim_y_neg = cam.cy / cam.fy + 0.3f * tan_fovy;
        const float rz = 1.0f / z, rz2 = rz * rz, rz3 = rz2 * rz;
        const float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, x * rz));
        const float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, y * rz));
        const float J[6] = { cam.fx * rz, 0.0f, -cam.fx * tx * rz2, 0.0f, cam.fy * rz, -cam.fy * ty * rz2 };
        float GJ[6], v_Cc[9], JC[6], JCt[6], v_J[6];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) GJ[i * 3 + j] = G[i * 2 + 0] * J[j] + G[i * 2 + 1] * J[3 + j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) v_Cc[i * 3 + j] = J[i] * GJ[j] + J[3 + i] * GJ[3 + j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                JC[i * 3 + j]  = J[i * 3 + 0] * Cc[0 * 3 + j] + J[i * 3 + 1] * Cc[1 * 3 + j] + J[i * 3 + 2] * Cc[2 * 3 + j];
                JCt[i * 3 + j] = J[i * 3 + 0] * Cc[j * 3 + 0] + J[i * 3 + 1] * Cc[j * 3 + 1] + J[i * 3 + 2] * Cc[j * 3 + 2];
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                v_J[i * 3 + j] = (G[i * 2 + 0] * JCt[j] + G[i * 2 + 1] * JCt[3 + j]) + (G[i] * JC[j] + G[2 + i] * JC[3 + j]);
        const float vm2x = rec[0], vm2y = rec[1];
        float v_mc[3];
        v_mc[0] = cam.fx * rz * vm2x;
        v_mc[1] = cam.fy * rz * vm2y;
        v_mc[2] = -(cam.fx * x * vm2x + cam.fy * y * vm2y) * rz2;
        if (x * rz <= lim_x_pos && x * rz >= -lim_x_neg) v_mc[0] += -cam.fx * rz2 * v_J[2];
        else                                             v_mc[2] += -cam.fx * rz3 * v_J[2] * tx;
        if (y * rz <= lim_y_pos && y * rz >= -lim_y_neg) v_mc[1] += -cam.fy * rz2 * v_J[5];
        else                                             v_mc[2] += -cam.fy * rz3 * v_J[5] * ty;
        v_mc[2] += -cam.fx * rz2 * v_J[0] - cam.fy * rz2 * v_J[4] + 2.0f * cam.fx * tx * rz3 * v_J[2]
                 + 2.0f * cam.fy * ty * rz3 * v_J[5];
        if (v_depths) v_mc[2] += v_depths[v];

        const float* R = cam.R;
#pragma unroll
        for (int i = 0; i < 3; ++i) g_mean[i] = R[0 * 3 + i] * v_mc[0] + R[1 * 3 + i] * v_mc[1] + R[2 * 3 + i] * v_mc[2];
        float Tm[9], v_cov[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                Tm[i * 3 + j] = R[0 * 3 + i] * v_Cc[0 * 3 + j] + R[1 * 3 + i] * v_Cc[1 * 3 + j] + R[2 * 3 + i] * v_Cc[2 * 3 + j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                v_cov[i * 3 + j] = Tm[i * 3 + 0] * R[0 * 3 + j] + Tm[i * 3 + 1] * R[1 * 3 + j] + Tm[i * 3 + 2] * R[2 * 3 + j];
        float v_M[9], v_Rq[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < 3; ++k) acc += (v_cov[i * 3 + k] + v_cov[k * 3 + i]) * M[k * 3 + j];
                v_M[i * 3 + j] = acc;
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) v_Rq[i * 3 + j] = v_M[i * 3 + j] * scale[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) g_scale[j] = Rq[0 * 3 + j] * v_M[0 * 3 + j] + Rq[1 * 3 + j] * v_M[1 * 3 + j] + Rq[2 * 3 + j] * v_M[2 * 3 + j];

        const float inv = 1.0f / sqrtf(q4.y * q4.y + q4.z * q4.z + q4.w * q4.w + q4.x * q4.x);
        const float w = q4.x * inv, xq = q4.y * inv, yq = q4.z * inv, zq = q4.w * inv;
#define VR(i, j) v_Rq[(i) * 3 + (j)]
        float vqn[4];
        vqn[0] = 2.0f * (xq * (VR(2, 1) - VR(1, 2)) + yq * (VR(0, 2) - VR(2, 0)) + zq * (VR(1, 0) - VR(0, 1)));
        vqn[1] = 2.0f * (-2.0f * xq * (VR(1, 1) + VR(2, 2)) + yq * (VR(1, 0) + VR(0, 1)) + zq * (VR(2, 0) + VR(0, 2)) + w * (VR(2, 1) - VR(1, 2)));
        vqn[2] = 2.0f * (xq * (VR(1, 0) + VR(0, 1)) - 2.0f * yq * (VR(0, 0) + VR(2, 2)) + zq * (VR(2, 1) + VR(1, 2)) + w * (VR(0, 2) - VR(2, 0)));
        vqn[3] = 2.0f * (xq * (VR(2, 0) + VR(0, 2)) + yq * (VR(2, 1) + VR(1, 2)) - 2.0f * zq * (VR(0, 0) + VR(1, 1)) + w * (VR(1, 0) - VR(0, 1)));
#undef VR
        const float qn[4] = { w, xq, yq, zq };
        const float dotp = vqn[0] * qn[0] + vqn[1] * qn[1] + vqn[2] * qn[2] + vqn[3] * qn[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) g_quat[k] = (vqn[k] - dotp * qn[k]) * inv;
    }
    if (accumulate) {          // += into persistent gradient buffers (several views per step); culled Gaussians add nothing
        if (v >= 0) {
            v_means[3 * (size_t)n] += g_mean[0]; v_means[3 * (size_t)n + 1] += g_mean[1]; v_means[3 * (size_t)n + 2] += g_mean[2];
            float4 q = *reinterpret_cast<float4*>(v_quats + 4 * (size_t)n);
            q.x += g_quat[0]; q.y += g_quat[1]; q.z += g_quat[2]; q.w += g_quat[3];
            *reinterpret_cast<float4*>(v_quats + 4 * (size_t)n) = q;
            v_scales[3 * (size_t)n] += g_scale[0]; v_scales[3 * (size_t)n + 1] += g_scale[1]; v_scales[3 * (size_t)n + 2] += g_scale[2];
            v_opacities[n] += g_op;
        }
    } else {
        v_means[3 * (size_t)n] = g_mean[0]; v_means[3 * (size_t)n + 1] = g_mean[1]; v_means[3 * (size_t)n + 2] = g_mean[2];
        *reinterpret_cast<float4*>(v_quats + 4 * (size_t)n) = make_float4(g_quat[0], g_quat[1], g_quat[2], g_quat[3]);
        v_scales[3 * (size_t)n] = g_scale[0]; v_scales[3 * (size_t)n + 1] = g_scale[1]; v_scales[3 * (size_t)n + 2] = g_scale[2];
        v_opacities[n] = g_op;
    }
    if (v_colors) {             // per-view quantity (feeds the shading backward): always a plain write
        for (int k = 0; k < D; ++k) v_colors[(size_t)n * D + k] = (v >= 0) ? v_packed[(size_t)v * rec_stride + 6 + k] : 0.0f;
    }
}

static int project_bwd_impl(int N, GsCount vc, int D, const float* means, const float* quats, const float* scales,
                            const float* opacities, const float* viewmat, const float* K, int W, int H,
                            float eps2d, const int32_t* gaussian_ids, const float* conics,
                            const float* compensations, const float* v_packed, int rec_stride,
                            const float* v_depths, float* v_means, float* v_quats, float* v_scales, float* v_opacities, float* v_colors,
                            int accumulate, void* stream)
{
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(project_bwd_kernel, dim3(gs_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, N, vc, D, means,
                       quats, scales, opacities, viewmat, K, W, H, eps2d, gaussian_ids, conics, compensations,
                       v_packed, rec_stride > 0 ? rec_stride : ((6 + D) + 15) / 16 * 16, v_depths, v_means, v_quats,
                       v_scales, v_opacities, v_colors, accumulate);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_project_bwd(int N, int V, int D, const float* means, const float* quats, const float* scales,
                              const float* opacities, const float* viewmat, const float* K, int W, int H,
                              float eps2d, const int32_t* gaussian_ids, const float* conics,
                              const float* compensations, const float* v_packed, int rec_stride,
                              const float* v_depths, float* v_means, float* v_quats, float* v_scales, float* v_opacities, float* v_colors,
                              int accumulate, void* stream)
{
    GS_CHECK_ARG(N >= 0 && V >= 0 && V <= N, "bad sizes");
    return project_bwd_impl(N, GsCount{ V, nullptr }, D, means, quats, scales, opacities, viewmat, K, W, H, eps2d, gaussian_ids, conics,
                            compensations, v_packed, rec_stride, v_depths, v_means, v_quats, v_scales, v_opacities, v_colors, accumulate,
                            stream);
}

extern "C" int gs_project_bwd_cap(int N, const int64_t* counts_dev, int D, const float* means, const float* quats, const float* scales,
                                  const float* opacities, const float* viewmat, const float* K, int W, int H,
                                  float eps2d, const int32_t* gaussian_ids, const float* conics,
                                  const float* compensations, const float* v_packed, int rec_stride,
                                  const float* v_depths, float* v_means, float* v_quats, float* v_scales, float* v_opacities,
                                  float* v_colors, int accumulate, void* stream)
{
    GS_CHECK_ARG(N >= 0 && counts_dev != nullptr, "bad sizes");
    return project_bwd_impl(N, GsCount{ N, (const long long*)counts_dev }, D, means, quats, scales, opacities, viewmat, K, W, H, eps2d,
                            gaussian_ids, conics, compensations, v_packed, rec_stride, v_depths, v_means, v_quats, v_scales, v_opacities,
                            v_colors, accumulate, stream);
}
