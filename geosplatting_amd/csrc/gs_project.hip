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

#include "gs_project_dev.h"
#pragma clang fp contract(off)


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
    i_incl = gs_wave_incl_scan(i_incl);
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
        const float* __restrict__ rec = v_packed + (size_t)v * rec_stride;   // {xy(2), conic(3), opacity, colors(D)}
        ProjGrad pg;
        project_bwd_one(cam, mean, q4, scale, opacities[n], (float)W, (float)H, eps2d, conics[3 * (size_t)v], conics[3 * (size_t)v + 1],
                        conics[3 * (size_t)v + 2], compensations[v], rec[0], rec[1], rec[2], rec[3], rec[4], rec[5],
                        v_depths ? v_depths[v] : 0.0f, pg);
#pragma unroll
        for (int k = 0; k < 3; ++k) { g_mean[k] = pg.mean[k]; g_scale[k] = pg.scale[k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) g_quat[k] = pg.quat[k];
        g_op = pg.op;
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
