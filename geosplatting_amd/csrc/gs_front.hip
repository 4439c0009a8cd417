// gs_front.hip -- the engine's fused per-Gaussian kernels (round 4): shading fused AHEAD of the projection, and the projection
// backward fused with the shading backward.
//
// The reference shades every Gaussian (rfstudio/model/geosplat.py:80-122: ~25 elementwise launches + two texture launches), hands
// the colours to gsplat.rasterization (rfstudio/model/gsplat.py:334-355), and autograd runs the two backwards one after the other.
// Rounds 1-3 kept that cut: gs_shade_fwd -> colours [N,3] -> gs_project_fwd_vis, and gs_project_bwd -> v_colors [N,3] ->
// gs_shade_bwd -- two N-sized round trips, four launches and two extra reads of the means per view.  Here:
//   front_fwd_kernel : one thread per Gaussian projects it (canonical order: project_exact), shades it if it is visible
//                      (shade_one: FG LUT, trilinear cube taps) and writes ONE 64-byte per-visible record -- the compositor's record
//                      with the colour in it, plus {compensation, Gaussian index, depth, radius} in what was padding -- together with
//                      the binning's inputs (a 24- or 32-bit depth key and the packed tile rectangle).  Nothing else is written: the
//                      separate gaussian_ids / radii / means2d / depths / conics / compensations / opacities / colours /
//                      tiles_per_gauss / cum_tiles arrays of the rasterization() call shape (68 B per visible Gaussian) do not exist on
//                      this path.  Packed order = ascending Gaussian index, by the same in-launch chained scan as project_fwd_kernel;
//                      the chunk aggregate is published BEFORE the texture taps, so that successors' look-backs overlap them.
//   tail_bwd_kernel  : one thread per VISIBLE Gaussian (packed slot): reads its 64-byte record and its 64-byte gradient record,
//                      chains the compositor's gradients through the projection (project_bwd_one) and the colour gradient through
//                      the shading (recomputed forward), adds every parameter gradient ONCE (means get the sum of both paths) and
//                      scatters the texel gradients exactly as shade_bwd_kernel does (LDS copies of the <= 32^2 levels, row-pair
//                      atomics for the rest).  No slot search, no [N,3] colour gradient.
// Arithmetic is the shared device code of gs_project_dev.h / gs_shade_dev.h: records, keys and rectangles are bit-identical to
// what gs_shade_fwd + gs_project_fwd_vis + tile_rect_kernel produce (tests/test_gpu_front.py).
#include "gs_common.h"
#include <string.h>
#pragma clang fp contract(off)
#include "gs_project_dev.h"
#include "gs_shade_dev.h"

// max over the 64 lanes of a wave, valid in lane 63 (lanes without a value pass 0): DPP moves fused into the VALU max, no LDS
// crossbar (twelve ds_bpermute per wave as __shfl_xor)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v)
{
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32_lane63(unsigned v)
{
    v = dpp_max_u32<0xB1, 0xf>(v);     // quad_perm [1,0,3,2]
    v = dpp_max_u32<0x4E, 0xf>(v);     // quad_perm [2,3,0,1]
    v = dpp_max_u32<0x141, 0xf>(v);    // row_half_mirror
    v = dpp_max_u32<0x140, 0xf>(v);    // row_mirror: every lane of a 16-lane row holds the row's max
    v = dpp_max_u32<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
    v = dpp_max_u32<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3
    return v;
}

__global__ void __launch_bounds__(GS_PROJ_BLOCK)
front_fwd_kernel(int N, const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
                 const float* __restrict__ opacities, const float* __restrict__ normals, const float* __restrict__ kd,
                 const float* __restrict__ ks, const float* __restrict__ viewmat, const float* __restrict__ K,
                 const float* __restrict__ cam_pos, float min_roughness, float max_metallic, int mode, EnvDev env,
                 int W, int H, int tile_size, int tile_w, int tile_h, float eps2d, float near_plane, float far_plane, float radius_clip,
                 unsigned key_base, unsigned key_limit /* 0: 32-bit keys */, float4* __restrict__ vis, unsigned* __restrict__ depth_keys,
                 uint2* __restrict__ rects, unsigned* __restrict__ ctrl, u64* __restrict__ desc, int n_chunks,
                 unsigned long long* __restrict__ counts /* {V, I, max(~depth bits), max(depth bits)} */, long long* __restrict__ status,
                 unsigned* __restrict__ tile_counts /* [tile_w * tile_h] or NULL */, int32_t* __restrict__ packed_index /* [N] or NULL */,
                 int tight_tiles)
{
    // intersections per TILE, counted here -- in Gaussian-index order, where neighbours on the surface share their tiles (a block of
    // 512 Gaussians touches a few dozen tiles; in the depth order of the emission it touched ~1 900 of the 2 500, one atomic each) --
    // in an LDS histogram flushed with one atomic per touched tile: the binning turns the counters into the tile offsets
    extern __shared__ unsigned s_th[];
    const int n_tiles = tile_w * tile_h;
    if (tile_counts) for (int t = threadIdx.x; t < n_tiles; t += GS_PROJ_BLOCK) s_th[t] = 0u;
    __shared__ int s_chunk;
    __shared__ int s_wv[GS_PROJ_WAVES];
    __shared__ int s_wi[GS_PROJ_WAVES];
    __shared__ unsigned s_wmx[GS_PROJ_WAVES], s_wmn[GS_PROJ_WAVES];
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

    const GsCam cam = load_cam(viewmat, K);
    ProjFwd p;
    p.valid = false; p.radius = 0;
    int ntiles = 0;
    int tx0 = 0, ty0 = 0, tx1 = 0, ty1 = 0;
    float mean[3] = { 0.f, 0.f, 0.f };
    float op = 0.0f, hx = -1.0f, hy = -1.0f;
    if (n < N) {
        mean[0] = means[3 * (size_t)n]; mean[1] = means[3 * (size_t)n + 1]; mean[2] = means[3 * (size_t)n + 2];
        const float4 q = *reinterpret_cast<const float4*>(quats + 4 * (size_t)n);
        const float quat[4] = { q.x, q.y, q.z, q.w };
        const float scale[3] = { scales[3 * (size_t)n], scales[3 * (size_t)n + 1], scales[3 * (size_t)n + 2] };
        p = project_exact(cam, mean, quat, scale, (float)W, (float)H, eps2d, near_plane, far_plane, radius_clip);
        if (p.valid) {
            tile_range_exact(p.m2x, p.m2y, p.radius, tile_size, tile_w, tile_h, tx0, ty0, tx1, ty1);
            op = opacities[n] * p.comp;
            if (!alpha_extent(p.ca, p.cb, p.cc, op, hx, hy)) { hx = -1.0f; hy = -1.0f; }
            if (tight_tiles) {
                // gsplat's tile rectangle is the square of radius ceil(3 sigma_max); a pixel can only composite this Gaussian inside
                // the axis-aligned extent of {alpha >= 1/255} (alpha_extent: the SAME conservative extents the compositor culls its
                // quadrants with, so dropping the tiles outside them changes no pixel).  A caller that returns gsplat's `meta` keeps
                // the square; the engine does not: 22 % fewer intersections to sort, stream and cull on the bench scene.
                if (hx < 0.0f) { tx1 = tx0; ty1 = ty0; }                  // can never reach alpha_min: no tile at all
                else {
                    const float ts = (float)tile_size;
                    const float fx0 = floorf((p.m2x - hx) / ts), fy0 = floorf((p.m2y - hy) / ts);
                    const float fx1 = ceilf((p.m2x + hx) / ts), fy1 = ceilf((p.m2y + hy) / ts);
                    const int ax0 = fx0 < 0.0f ? 0 : (fx0 > (float)tile_w ? tile_w : (int)fx0), ax1 = fx1 < 0.0f ? 0 : (fx1 > (float)tile_w ? tile_w : (int)fx1);
                    const int ay0 = fy0 < 0.0f ? 0 : (fy0 > (float)tile_h ? tile_h : (int)fy0), ay1 = fy1 < 0.0f ? 0 : (fy1 > (float)tile_h ? tile_h : (int)fy1);
                    tx0 = max(tx0, ax0); tx1 = min(tx1, ax1); ty0 = max(ty0, ay0); ty1 = min(ty1, ay1);
                    if (tx1 < tx0) tx1 = tx0;
                    if (ty1 < ty0) ty1 = ty0;
                }
            }
            ntiles = (tx1 - tx0) * (ty1 - ty0);
        }
    }

    // ---- block-level scan of (valid, ntiles); the chunk's aggregate goes out at once
    const u64 bal = __ballot(p.valid);
    const int v_excl_wave = __popcll(bal & ((1ull << lane) - 1ull));
    int i_incl = ntiles;
    i_incl = gs_wave_incl_scan(i_incl);
    const unsigned dbits = p.valid ? __float_as_uint(p.depth) : 0u;
    {
        const unsigned mx = wave_max_u32_lane63(dbits), mn = wave_max_u32_lane63(p.valid ? ~dbits : 0u);
        if (lane == 63) { s_wv[wave] = __popcll(bal); s_wi[wave] = i_incl; s_wmx[wave] = mx; s_wmn[wave] = mn; }
    }
    __syncthreads();
    int v_before = 0, i_before = 0, aggV = 0, aggI = 0;
#pragma unroll
    for (int w = 0; w < GS_PROJ_WAVES; ++w) {
        if (w < wave) { v_before += s_wv[w]; i_before += s_wi[w]; }
        aggV += s_wv[w]; aggI += s_wi[w];
    }
    if (wave == 0 && lane == 0 && chunk > 0) { desc_store(&agg_v[chunk], (u64)aggV); desc_store(&agg_i[chunk], (u64)aggI); }

    const bool range_wave = wave == GS_PROJ_WAVES - 1;        // (not the first one: that one walks the look-back)
    unsigned long long cur_mx = 0ull, cur_mn = 0ull;
    if (range_wave && lane == 0 && aggV != 0) {
        cur_mx = __hip_atomic_load(counts + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur_mn = __hip_atomic_load(counts + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- shading of the visible Gaussians (texture taps in flight while the predecessors publish)
    float color[3] = { 0.f, 0.f, 0.f };
    if (p.valid && vis != nullptr) {                          // (vis == NULL: geometry only -- keys, rectangles, counts, slots)
        const float nrm[3] = { normals[3 * (size_t)n], normals[3 * (size_t)n + 1], normals[3 * (size_t)n + 2] };
        const float kdn[3] = { kd[3 * (size_t)n], kd[3 * (size_t)n + 1], kd[3 * (size_t)n + 2] };
        const float2 ks2 = *reinterpret_cast<const float2*>(ks + 2 * (size_t)n);
        const float ksn[2] = { ks2.x, ks2.y };
        const float cp[3] = { cam_pos[0], cam_pos[1], cam_pos[2] };
        ShadeTmp t;
        shade_one<false>(mean, nrm, kdn, ksn, cp, min_roughness, max_metallic, mode, env, color, t);
    }
    // depth range of the view (the engine sizes its 24-bit keys by it, one step later).  The maxima only grow, so a relaxed -- possibly
    // stale, never too large -- read of the current value decides whether this BLOCK can still raise it: a handful of atomics per
    // launch.  The reads are agent-scope (served past the CU's vector L1, which another CU's atomics never refresh), all of the same
    // L2 line: one wave per block issues them BEFORE its shading (cur_mx / cur_mn above) and uses them here.  Round 6 (removal
    // experiments, scripts/xp/front_alone.py): the per-wave version of this block -- twelve ds_bpermute for the two wave maxima, a
    // load pair per wave consumed at once -- cost 14 us of the kernel's 197 alone; per block, with DPP maxima and DPP scans: 187.
    // [One atomic pair per wave, unconditionally: 730 us instead of 260.]
    if (range_wave && lane == 0 && aggV != 0) {
        unsigned mx = 0u, mn = 0u;
#pragma unroll
        for (int w = 0; w < GS_PROJ_WAVES; ++w) { mx = s_wmx[w] > mx ? s_wmx[w] : mx; mn = s_wmn[w] > mn ? s_wmn[w] : mn; }
        if ((unsigned long long)mx > cur_mx) atomicMax(counts + 3, (unsigned long long)mx);
        if ((unsigned long long)mn > cur_mn) atomicMax(counts + 2, (unsigned long long)mn);
    }
    unsigned key = dbits;
    if (key_limit != 0u) {
        const bool fits = dbits >= key_base && (dbits - key_base) < key_limit;
        if (p.valid && !fits && status) status[3] = 1;                        // (sticky; the engine then falls back to 32-bit keys)
        key = fits ? dbits - key_base : (dbits < key_base ? 0u : key_limit - 1u);
    }

    // ---- decoupled look-back across chunks (wave 0): see project_fwd_kernel
    if (wave == 0) {
        long long baseV = 0, baseI = 0;
        if (chunk > 0) {
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
                        if (++spins > GS_SPIN_LIMIT) {             // a predecessor never published: the packed slots behind it are wrong.
                            atomicExch(ctrl + 1, 1u);                 // Reported like a truncated view (memory-safe, wrong, repeat the step)
                            if (status) status[0] = GS_ENOSPC;
                            isA = true; av = ai = 0; break;
                        }
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
            if (chunk == n_chunks - 1) { counts[0] = (unsigned long long)(baseV + aggV); counts[1] = (unsigned long long)(baseI + aggI); }
        }
    }
    __syncthreads();
    if (packed_index && n < N) packed_index[n] = p.valid ? (int32_t)(s_base[0] + v_before + v_excl_wave) : -1;
    if (p.valid) {
        const long long slot = s_base[0] + v_before + v_excl_wave;
        if (vis != nullptr) {
            float4* rec = vis + 4 * slot;
            rec[0] = make_float4(p.m2x, p.m2y, 0.5f * p.ca, p.cb);
            rec[1] = make_float4(0.5f * p.cc, op, hx, hy);
            rec[2] = make_float4(color[0], color[1], color[2], 0.0f);
            rec[3] = make_float4(p.comp, __int_as_float(n), p.depth, __int_as_float(p.radius));
        }
        if (depth_keys != nullptr) {
            depth_keys[slot] = key;
            rects[slot] = make_uint2((unsigned)tx0 | ((unsigned)ty0 << 16), (unsigned)tx1 | ((unsigned)ty1 << 16));
        }
        if (tile_counts)
            for (int i = ty0; i < ty1; ++i)
                for (int j = tx0; j < tx1; ++j) atomicAdd(&s_th[i * tile_w + j], 1u);
    }
    if (tile_counts) {
        __syncthreads();
        for (int t = threadIdx.x; t < n_tiles; t += GS_PROJ_BLOCK) { const unsigned c = s_th[t]; if (c) atomicAdd(tile_counts + t, c); }
    }
}

// one launch clears the chained-scan state, the counts and the tile counters
__global__ void __launch_bounds__(256)
front_setup_kernel(unsigned* __restrict__ ws, int ws_words, unsigned long long* __restrict__ counts4, unsigned* __restrict__ tile_counts, int n_tiles)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (int k = i; k < ws_words; k += stride) ws[k] = 0u;
    for (int k = i; k < n_tiles; k += stride) tile_counts[k] = 0u;
    if (i < 4) counts4[i] = 0ull;
}

#define GS_FRONT_HIST_MAX 8192               // tiles whose counters fit a block's LDS histogram (32 KB); gs_isect_bin_front agrees
extern "C" size_t gs_front_ws_bytes(int N) { return (gs_project_ws_bytes(N) + 15) & ~(size_t)15; }

extern "C" int gs_front_fwd(int N, const float* means, const float* quats, const float* scales, const float* opacities,
                            const float* normals, const float* kd, const float* ks, const float* viewmat, const float* K,
                            const float* cam_pos, float min_roughness, float max_metallic, int mode, const GsEnv* env,
                            int W, int H, int tile_size, float eps2d, float near_plane, float far_plane, float radius_clip,
                            uint32_t key_base, int key_bits, float* vis_records, uint32_t* depth_keys, uint32_t* tile_rects,
                            uint32_t* tile_counts, int32_t* packed_index, int tight_tiles, int64_t* counts4, int64_t* status4, void* ws,
                            size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(N >= 0 && W > 0 && H > 0 && tile_size > 0 && mode >= 0 && mode <= 2, "bad sizes or mode");
    GS_CHECK_ARG(counts4 != nullptr && ws != nullptr, "null argument");
    GS_CHECK_ARG((depth_keys != nullptr) == (tile_rects != nullptr) && (vis_records != nullptr || depth_keys != nullptr),
                 "depth_keys and tile_rects go together; at least one of {vis_records} / {depth_keys, tile_rects} is wanted");
    GS_CHECK_ARG(key_bits == 24 || key_bits == 32, "key_bits must be 24 or 32");
    GS_CHECK_ARG(key_bits == 32 || status4 != nullptr, "24-bit keys need a status word (range overflow is reported there)");
    EnvDev e;
    if (vis_records != nullptr) GS_CHECK_ARG(env_to_dev(env, e) == 0, "bad GsEnv");
    else memset(&e, 0, sizeof(e));                            // geometry only: nothing is shaded
    if (ws_bytes < gs_front_ws_bytes(N)) { gs_set_error("gs_front_fwd: workspace too small"); return GS_ENOSPC; }
    hipStream_t s = (hipStream_t)stream;
    const int tile_w = (W + tile_size - 1) / tile_size, tile_h = (H + tile_size - 1) / tile_size;
    GS_CHECK_ARG(tile_w < 65536 && tile_h < 65536, "tile grid too large");
    const int n_tiles = tile_w * tile_h;
    const bool hist = tile_counts != nullptr && n_tiles <= GS_FRONT_HIST_MAX;          // more tiles: the binning derives the offsets itself
    const int ws_words = (int)(gs_front_ws_bytes(N) / 4);
    hipLaunchKernelGGL(front_setup_kernel, dim3(gs_cdiv(ws_words > n_tiles ? ws_words : n_tiles, 256) < 64 ? gs_cdiv(ws_words > n_tiles ? ws_words : n_tiles, 256) : 64),
                       dim3(256), 0, s, (unsigned*)ws, ws_words, (unsigned long long*)counts4, tile_counts, tile_counts ? n_tiles : 0);
    GS_CHECK_LAUNCH();
    if (N == 0) return GS_OK;
    const int n_chunks = (N + GS_PROJ_BLOCK - 1) / GS_PROJ_BLOCK;
    hipLaunchKernelGGL(front_fwd_kernel, dim3(n_chunks), dim3(GS_PROJ_BLOCK), hist ? (size_t)n_tiles * sizeof(unsigned) : 0, s, N, means, quats, scales, opacities, normals, kd, ks,
                       viewmat, K, cam_pos, min_roughness, max_metallic, mode, e, W, H, tile_size, tile_w, tile_h, eps2d, near_plane,
                       far_plane, radius_clip, key_bits == 24 ? key_base : 0u, key_bits == 24 ? (1u << 24) : 0u, (float4*)vis_records,
                       depth_keys, (uint2*)tile_rects, (unsigned*)ws, (u64*)((char*)ws + 16), n_chunks, (unsigned long long*)counts4,
                       (long long*)status4, hist ? tile_counts : (unsigned*)nullptr, packed_index, tight_tiles);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// tail: A7 + S1-S3 backward of one view, accumulated into the caller's gradient buffers (always +=)
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
tail_bwd_kernel(GsCount vc, const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
                const float* __restrict__ opacities, const float* __restrict__ normals, const float* __restrict__ kd,
                const float* __restrict__ ks, const float* __restrict__ viewmat, const float* __restrict__ K,
                const float* __restrict__ cam_pos, float min_roughness, float max_metallic, int mode, EnvDev env, int W, int H,
                float eps2d, const float4* __restrict__ vis, const float* __restrict__ v_packed, int rec_stride,
                float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales, float* __restrict__ v_opacities,
                float* __restrict__ v_normals, float* __restrict__ v_kd, float* __restrict__ v_ks, EnvGradDev eg)
{
    extern __shared__ __attribute__((aligned(16))) float s_grad[];
    for (int i = threadIdx.x; i < eg.lds_floats; i += blockDim.x) s_grad[i] = 0.0f;
    __syncthreads();
    float* const stage = s_grad + eg.stage_off + (threadIdx.x >> 6) * 640;            // (the tail always stages its row-pair commits)
    // this XCD's private copy of the mid-sized levels (workgroup-scope atomics resolve in the XCD's own L2), or nullptr
    float* const priv = eg.priv ? eg.priv + (long long)gs_xcc_id() * eg.priv_stride : nullptr;
    const float cp[3] = { cam_pos[0], cam_pos[1], cam_pos[2] };
    const GsCam cam = load_cam(viewmat, K);
    const int V = (int)gs_count(vc);
    const int stride = (int)(gridDim.x * blockDim.x);
    const int n_iter = (V + stride - 1) / stride;                   // wave-uniform: every lane reaches the wave-aggregated scatter
    for (int it = 0; it < n_iter; ++it) {
        const int v = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool scatter = false;
        ShadeTmp t;
        float v_ls[3] = { 0, 0, 0 }, v_ld[3] = { 0, 0, 0 };
        if (v < V) {
            const float4 r0 = vis[4 * (size_t)v], r1 = vis[4 * (size_t)v + 1], r3 = vis[4 * (size_t)v + 3];
            const int n = __float_as_int(r3.y);
            const float4* gp = reinterpret_cast<const float4*>(v_packed + (size_t)v * rec_stride);   // {xy(2), conic(3), opacity, rgb}
            const float4 g0 = gp[0], g1 = gp[1];
            const float g2 = v_packed[(size_t)v * rec_stride + 8];
            const float g[3] = { g1.z, g1.w, g2 };
            const bool any_geo = g0.x != 0.0f || g0.y != 0.0f || g0.z != 0.0f || g0.w != 0.0f || g1.x != 0.0f || g1.y != 0.0f;
            const bool any_col = g[0] != 0.0f || g[1] != 0.0f || g[2] != 0.0f;
            if (any_geo || any_col) {                                  // (a Gaussian that reached no pixel: every gradient is exactly zero)
                const float mean[3] = { means[3 * (size_t)n], means[3 * (size_t)n + 1], means[3 * (size_t)n + 2] };
                float o_mean[3] = { 0, 0, 0 };
                if (any_geo) {
                    const float4 q4 = *reinterpret_cast<const float4*>(quats + 4 * (size_t)n);
                    const float scale[3] = { scales[3 * (size_t)n], scales[3 * (size_t)n + 1], scales[3 * (size_t)n + 2] };
                    ProjGrad pg;
                    project_bwd_one(cam, mean, q4, scale, opacities[n], (float)W, (float)H, eps2d, 2.0f * r0.z, r0.w, 2.0f * r1.x, r3.x,
                                    g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, 0.0f, pg);
                    float4 q = *reinterpret_cast<float4*>(v_quats + 4 * (size_t)n);
                    q.x += pg.quat[0]; q.y += pg.quat[1]; q.z += pg.quat[2]; q.w += pg.quat[3];
                    *reinterpret_cast<float4*>(v_quats + 4 * (size_t)n) = q;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { v_scales[3 * (size_t)n + k] += pg.scale[k]; o_mean[k] = pg.mean[k]; }
                    v_opacities[n] += pg.op;
                }
                if (any_col) {
                    const float normal[3] = { normals[3 * (size_t)n], normals[3 * (size_t)n + 1], normals[3 * (size_t)n + 2] };
                    const float kdn[3] = { kd[3 * (size_t)n], kd[3 * (size_t)n + 1], kd[3 * (size_t)n + 2] };
                    const float2 ks2 = *reinterpret_cast<const float2*>(ks + 2 * (size_t)n);
                    const float ksn[2] = { ks2.x, ks2.y };
                    float color[3];
                    shade_one<true>(mean, normal, kdn, ksn, cp, min_roughness, max_metallic, mode, env, color, t);
                    scatter = true;
                    // ---- the arithmetic of shade_bwd_kernel (gs_shade.hip), same order
                    float v_diff[3] = { 0, 0, 0 }, v_rf[3] = { 0, 0, 0 };
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (mode == GS_MODE_PBR)          { v_diff[c] = g[c]; v_ls[c] = g[c] * t.refl_c[c]; v_rf[c] = g[c] * t.ls.out[c]; }
                        else if (mode == GS_MODE_DIFFUSE) { v_ld[c] = g[c] * t.diff[c]; v_diff[c] = g[c] * t.ld[c]; }
                        else                              { v_ls[c] = g[c] * t.refl_c[c]; v_rf[c] = g[c] * t.ls.out[c]; }
                    }
                    float v_A = 0.0f, v_B = 0.0f, v_metal = 0.0f, o_kd[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v_spec = v_rf[c] * t.fg[0];
                        v_A += v_rf[c] * t.spec[c];
                        v_B += v_rf[c];
                        o_kd[c] = v_spec * t.metal + v_diff[c] * (1.0f - t.metal);
                        v_metal += v_spec * (kdn[c] - 0.04f) - v_diff[c] * kdn[c];
                    }
                    const float v_ndv = v_A * t.dfg_du[0] + v_B * t.dfg_du[1];
                    float v_rough = v_A * t.dfg_dv[0] + v_B * t.dfg_dv[1];
                    float v_mip = 0.0f, v_refl[3] = { 0, 0, 0 }, v_n[3] = { 0, 0, 0 };
                    if (mode != GS_MODE_DIFFUSE) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            v_mip += v_ls[c] * t.ls.dmip[c];
#pragma unroll
                            for (int k = 0; k < 3; ++k) v_refl[k] += v_ls[c] * t.ls.dd[c * 3 + k];
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c)
#pragma unroll
                            for (int k = 0; k < 3; ++k) v_n[k] += v_ld[c] * t.ld_dd[c * 3 + k];
                    }
                    v_rough += v_mip * t.dmip_dr;
                    float v_d = 2.0f * (v_refl[0] * normal[0] + v_refl[1] * normal[1] + v_refl[2] * normal[2]);
                    float v_wo[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) { v_n[k] += 2.0f * t.d * v_refl[k]; v_wo[k] = -v_refl[k]; }
                    if (t.d >= 1e-6f) v_d += v_ndv;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { v_n[k] += v_d * t.wo[k]; v_wo[k] += v_d * normal[k]; }
                    if (!t.wo_const) {
                        const float dot = t.wo[0] * v_wo[0] + t.wo[1] * v_wo[1] + t.wo[2] * v_wo[2];
                        const float l = fmaxf(t.len, 1e-6f);
#pragma unroll
                        for (int k = 0; k < 3; ++k) o_mean[k] += -((v_wo[k] - t.wo[k] * dot) / l);
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) { v_normals[3 * (size_t)n + k] += v_n[k]; v_kd[3 * (size_t)n + k] += o_kd[k]; }
                    float2 o = *reinterpret_cast<float2*>(v_ks + 2 * (size_t)n);
                    o.x += v_rough * (1.0f - min_roughness); o.y += v_metal * max_metallic;
                    *reinterpret_cast<float2*>(v_ks + 2 * (size_t)n) = o;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) v_means[3 * (size_t)n + k] += o_mean[k];
            }
        }
        // ---- texel gradients: LDS-private copies for the small levels, row-pair atomics for the rest (as shade_bwd_kernel)
        if (mode != GS_MODE_DIFFUSE) {
            const int l0 = scatter ? t.ls.l0 : 0, l1 = scatter ? t.ls.l1 : -1;
            const float w0 = (l1 < 0) ? 1.0f : 1.0f - t.ls.f;
            const bool lds0 = scatter && eg.lds_level[l0] >= 0;
            if (lds0) cube_scatter_lds(s_grad + eg.lds_level[l0], t.ls.fp0, v_ls, w0);
            const bool loc0 = priv != nullptr && eg.priv_level[l0] >= 0;
            cube_scatter_wave_tagged(scatter && !lds0 ? (loc0 ? priv + eg.priv_level[l0] : eg.levels[l0]) : nullptr, loc0, t.ls.fp0, v_ls, w0,
                                     scatter && !lds0, stage);
            const bool has1 = scatter && l1 >= 0;
            const int l1s = has1 ? l1 : 0;
            const bool lds1 = has1 && eg.lds_level[l1s] >= 0;
            if (lds1) cube_scatter_lds(s_grad + eg.lds_level[l1], t.ls.fp1, v_ls, t.ls.f);
            const bool loc1 = priv != nullptr && eg.priv_level[l1s] >= 0;
            cube_scatter_wave_tagged(has1 && !lds1 ? (loc1 ? priv + eg.priv_level[l1s] : eg.levels[l1s]) : nullptr, loc1, t.ls.fp1, v_ls, t.ls.f,
                                     has1 && !lds1, stage);
        } else {
            const bool ldsb = scatter && eg.lds_base >= 0;
            if (ldsb) cube_scatter_lds(s_grad + eg.lds_base, t.ld_fp, v_ld, 1.0f);
            cube_scatter_wave_tagged(scatter && !ldsb ? eg.base : nullptr, false, t.ld_fp, v_ld, 1.0f, scatter && !ldsb, stage);
        }
    }
    // ---- flush the private copies
    __syncthreads();
    if (eg.lds_base >= 0) {
        const int cnt = 18 * env.base_res * env.base_res;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const float x = s_grad[eg.lds_base + i];
            if (x != 0.0f) gs_atomic_add(eg.base + i, x);
        }
    }
    for (int l = 0; l < env.L; ++l) {
        if (eg.lds_level[l] < 0) continue;
        const int cnt = 18 * env.res[l] * env.res[l];
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const float x = s_grad[eg.lds_level[l] + i];
            if (x != 0.0f) gs_atomic_add(eg.levels[l] + i, x);
        }
    }
}

// XCD-private copies of the mid-sized specular levels (not in LDS, at most GS_TAIL_PRIV_MAXRES^2 texels per face): float offsets
// persistent blocks of a BACKGROUND tail launch (bit 2 of `parts`): a launch that is not the last of the step runs beside the
// compositor of the following views; its 134 KB blocks cannot share a CU with the compositor's, so it takes half of the CUs and
// leaves the others (measured at 8 views, batches of 3: 96 / 128 / 160 / 192 / 256 blocks -> 669 / 678 / 675 / 675 / 652 views/s)
static int tail_background_blocks() { return 128; }            // workgroups of a background tail launch: half of the CUs (96 / 160 / 192 / 256: 669 / 675 / 675 / 652 against 678 views/s)
static int tail_priv_maxres() { return 128; }
static size_t tail_priv_floats(const EnvDev& e, int mode, long long* level_off)
{
    long long off = 0;
    for (int l = 0; l < GS_MAX_LEVELS; ++l) level_off[l] = -1;
    if (mode != GS_MODE_DIFFUSE)
        for (int l = 0; l < e.L; ++l)
            if (e.res[l] > 32 && e.res[l] <= tail_priv_maxres()) { level_off[l] = off; off += 18ll * e.res[l] * e.res[l]; }
    return (size_t)((off + 63) / 64 * 64);
}

extern "C" size_t gs_tail_priv_ws_bytes(const GsEnv* env, int mode)
{
    EnvDev e;
    if (env_to_dev(env, e) != 0) return 0;
    long long lo[GS_MAX_LEVELS];
    return tail_priv_floats(e, mode, lo) * sizeof(float) * GS_XCD_COPIES;
}

// env_grad.levels[l] += sum of the eight private copies, for the levels that have them (once per step, after the last gs_tail_bwd)
extern "C" int gs_tail_priv_reduce(const GsEnv* env, int mode, const void* priv_ws, size_t priv_ws_bytes, const GsEnvGrad* env_grad, void* stream)
{
    EnvDev e;
    GS_CHECK_ARG(env_to_dev(env, e) == 0 && env_grad != nullptr, "bad GsEnv / env_grad");
    long long lo[GS_MAX_LEVELS];
    const size_t floats = tail_priv_floats(e, mode, lo);
    if (floats == 0 || priv_ws == nullptr) return GS_OK;
    if (priv_ws_bytes < floats * sizeof(float) * GS_XCD_COPIES) { gs_set_error("gs_tail_priv_reduce: workspace too small"); return GS_ENOSPC; }
    for (int l = 0; l < e.L; ++l) {
        if (lo[l] < 0) continue;
        GS_CHECK_ARG(env_grad->levels[l] != nullptr, "env_grad->levels[l] required");
        const long long n = 18ll * e.res[l] * e.res[l];
        hipLaunchKernelGGL(priv_reduce_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream, n,
                           GS_XCD_COPIES, (const float*)priv_ws + lo[l], (long long)floats, env_grad->levels[l]);
        GS_CHECK_LAUNCH();
    }
    return GS_OK;
}

extern "C" int gs_tail_bwd(int V_cap, const int64_t* counts_dev /* NULL: V_cap is the count */, const float* means, const float* quats,
                           const float* scales, const float* opacities, const float* normals, const float* kd, const float* ks,
                           const float* viewmat, const float* K, const float* cam_pos, float min_roughness, float max_metallic, int mode,
                           const GsEnv* env, int W, int H, float eps2d, const float* vis_records, const float* v_packed, int rec_stride,
                           float* v_means, float* v_quats, float* v_scales, float* v_opacities, float* v_normals, float* v_kd, float* v_ks,
                           const GsEnvGrad* env_grad, void* priv_ws, size_t priv_ws_bytes, void* stream)
{
    GS_CHECK_ARG(V_cap >= 0 && mode >= 0 && mode <= 2 && rec_stride >= 12 && (rec_stride % 4) == 0, "bad sizes, mode or record stride");
    EnvDev e;
    GS_CHECK_ARG(env_to_dev(env, e) == 0, "bad GsEnv");
    GS_CHECK_ARG(env_grad != nullptr && vis_records != nullptr && v_packed != nullptr, "null argument");
    EnvGradDev eg;
    eg.base = env_grad->base;
    for (int l = 0; l < GS_MAX_LEVELS; ++l) eg.levels[l] = l < e.L ? env_grad->levels[l] : nullptr;
    if (mode == GS_MODE_DIFFUSE) GS_CHECK_ARG(eg.base != nullptr, "env_grad->base required in diffuse mode");
    else for (int l = 0; l < e.L; ++l) GS_CHECK_ARG(eg.levels[l] != nullptr, "env_grad->levels[l] required");
    ShadeBwdPlan plan;
    { const int rc = shade_bwd_plan(e, mode, V_cap, nullptr, 0, eg, plan, true); if (rc != GS_OK) return rc; }
    // optional XCD-private copies of the mid-sized levels (caller-zeroed, accumulated over the views of a step, folded by
    // gs_tail_priv_reduce)
    {
        const size_t floats = tail_priv_floats(e, mode, eg.priv_level);
        if (priv_ws != nullptr && floats > 0) {
            if (priv_ws_bytes < floats * sizeof(float) * GS_XCD_COPIES) { gs_set_error("gs_tail_bwd: private workspace too small"); return GS_ENOSPC; }
            eg.priv = (float*)priv_ws; eg.priv_stride = (long long)floats;
        } else {
            eg.priv = nullptr; eg.priv_stride = 0;
            for (int l = 0; l < GS_MAX_LEVELS; ++l) eg.priv_level[l] = -1;
        }
    }
    if (V_cap == 0) return GS_OK;
    hipStream_t s = (hipStream_t)stream;
    const GsCount vc{ (long long)V_cap, (const long long*)counts_dev };
#define GS_TAIL_LAUNCH(B)                                                                                                              \
    do {                                                                                                                               \
        GS_CHECK_HIP(hipFuncSetAttribute((const void*)tail_bwd_kernel<B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes)); \
        hipLaunchKernelGGL((tail_bwd_kernel<B>), dim3(plan.blocks), dim3(B), plan.lds_bytes, s, vc, means, quats, scales, opacities, normals, \
                           kd, ks, viewmat, K, cam_pos, min_roughness, max_metallic, mode, e, W, H, eps2d, (const float4*)vis_records,   \
                           v_packed, rec_stride, v_means, v_quats, v_scales, v_opacities, v_normals, v_kd, v_ks, eg);                    \
        GS_CHECK_LAUNCH();                                                                                                             \
    } while (0)
    if (plan.block == 1024) GS_TAIL_LAUNCH(1024); else if (plan.block == 768) GS_TAIL_LAUNCH(768); else if (plan.block == 256) GS_TAIL_LAUNCH(256);
    else GS_TAIL_LAUNCH(512);
#undef GS_TAIL_LAUNCH
    return GS_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// the tails of ALL views of a step (round 4)
#define GS_TAIL_MAX_VIEWS 8
struct TailViewDev {
    const float* viewmat; const float* K; const float* cam_pos;
    const float4* vis; const float* v_packed; const int32_t* packed_index;
    int W, H;
};
struct TailViewsDev { TailViewDev v[GS_TAIL_MAX_VIEWS]; int n; };

// The per-view tail reads and writes every parameter gradient once per view (76 B read + 76 B written per Gaussian and view: 43 % of
// its traffic); here the views of a Gaussian are handled together and its gradients stored once.  VIEWS ON THE LANES, TWO kernels:
// VPG (1, 2, 4 or 8) adjacent lanes own one Gaussian, one view each: a lane computes ONE (Gaussian, view) pair, the gradients are
// summed over the VPG lanes with DPP adds and stored once.  (First version, removed: one thread per Gaussian looping over the views --
// 19 accumulators and the temporaries of the projection AND the shading backward alive together, 256 VGPRs + scratch, two waves
// per SIMD behind dependent loads at 2 650 VALU instructions per pair: 2.9 ms per 8-view step against 2.6.)  Each half alone fits 128
// VGPRs -- fused, the register allocator overlaps them whatever the source order (250 VGPRs; 123 spilled at a 128 cap; a
// non-inlined call: 66) -- so they are two launches:
//   tail_shade_pairs_kernel : S1-S3 backward; pyramid levels one at a time, the colour cotangent contracted into the fetch
//                             (shade_pair_pre / cube_fetch_vjp / shade_pair_post, gs_shade_dev.h); sixteen waves per CU share the
//                             LDS texel copies; writes v_means (its view-direction part), v_normals, v_kd, v_ks
//   tail_proj_pairs_kernel  : A7; no LDS but the camera table; ADDS its part to v_means, writes v_quats, v_scales, v_opacities
// Price: the 64-byte gradient record and the packed slot of a pair are read by both (+1.1 GB per 8-view step).
#ifndef GS_TAILP_BLOCK
#define GS_TAILP_BLOCK 1024        // sixteen waves per CU on one set of LDS texel copies
#endif
struct TailViewLds {
    unsigned long long vis, v_packed, packed_index;
    float cam[16];                 // R[9], t[3], fx, fy, cx, cy
    float cam_pos[3];
    float Wf, Hf;
    float pad;
};

struct TailLevelLds { unsigned long long tex, gtex; long long priv_off; int R, lds_off; };   // priv_off / lds_off < 0: none

template <int VPG>
__device__ __forceinline__ float tail_group_sum(float v)
{
    if (VPG >= 2) v = gs_dpp_add<0xB1, 0xf, 0xf>(v);      // quad_perm [1,0,3,2]
    if (VPG >= 4) v = gs_dpp_add<0x4E, 0xf, 0xf>(v);      // quad_perm [2,3,0,1]
    if (VPG >= 8) v = gs_dpp_add<0x141, 0xf, 0xf>(v);     // row_half_mirror
    return v;
}

template <int BLOCK, int VPG>
__global__ void __launch_bounds__(BLOCK)
tail_proj_pairs_kernel(int N, TailViewsDev views, const float* __restrict__ means, const float* __restrict__ quats,
                       const float* __restrict__ scales, const float* __restrict__ opacities, float eps2d, int rec_stride,
                       float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
                       float* __restrict__ v_opacities, int accumulate)
{
    __shared__ TailViewLds s_views[VPG];
#pragma unroll
    for (int k = 0; k < VPG; ++k) {
        if ((int)threadIdx.x == k) {
            const TailViewDev& h = views.v[k];
            TailViewLds& d = s_views[k];
            d.vis = (unsigned long long)h.vis; d.v_packed = (unsigned long long)h.v_packed; d.packed_index = (unsigned long long)h.packed_index;
            const GsCam c = load_cam(h.viewmat, h.K);
#pragma unroll
            for (int i = 0; i < 9; ++i) d.cam[i] = c.R[i];
            d.cam[9] = c.t[0]; d.cam[10] = c.t[1]; d.cam[11] = c.t[2];
            d.cam[12] = c.fx; d.cam[13] = c.fy; d.cam[14] = c.cx; d.cam[15] = c.cy;
            d.cam_pos[0] = d.cam_pos[1] = d.cam_pos[2] = 0.0f;
            d.Wf = (float)h.W; d.Hf = (float)h.H; d.pad = 0.0f;
        }
    }
    __syncthreads();
    constexpr int G = BLOCK / VPG;
    const int view = (int)threadIdx.x & (VPG - 1), gl = (int)threadIdx.x / VPG;
    const TailViewLds& vw = s_views[view];
    const int n = (int)blockIdx.x * G + gl;
    const bool live = n < N && view < views.n;
    const int slot = live ? reinterpret_cast<const int32_t*>(vw.packed_index)[n] : -1;
    ProjGrad pg;
#pragma unroll
    for (int c = 0; c < 3; ++c) { pg.mean[c] = 0.0f; pg.scale[c] = 0.0f; }
#pragma unroll
    for (int c = 0; c < 4; ++c) pg.quat[c] = 0.0f;
    pg.op = 0.0f;
    if (slot >= 0) {
        const float* vp = reinterpret_cast<const float*>(vw.v_packed) + (size_t)slot * rec_stride;
        const float4 g0 = reinterpret_cast<const float4*>(vp)[0];
        const float2 g1 = reinterpret_cast<const float2*>(vp)[2];
        if (g0.x != 0.0f || g0.y != 0.0f || g0.z != 0.0f || g0.w != 0.0f || g1.x != 0.0f || g1.y != 0.0f) {
            const float4* vis = reinterpret_cast<const float4*>(vw.vis);
            const float4 r0 = vis[4 * (size_t)slot], r1 = vis[4 * (size_t)slot + 1], r3 = vis[4 * (size_t)slot + 3];
            const float mean[3] = { means[3 * (size_t)n], means[3 * (size_t)n + 1], means[3 * (size_t)n + 2] };
            const float scale[3] = { scales[3 * (size_t)n], scales[3 * (size_t)n + 1], scales[3 * (size_t)n + 2] };
            const float4 q4 = *reinterpret_cast<const float4*>(quats + 4 * (size_t)n);
            const float opac = opacities[n];
            GsCam cam;
#pragma unroll
            for (int i = 0; i < 9; ++i) cam.R[i] = vw.cam[i];
            cam.t[0] = vw.cam[9]; cam.t[1] = vw.cam[10]; cam.t[2] = vw.cam[11];
            cam.fx = vw.cam[12]; cam.fy = vw.cam[13]; cam.cx = vw.cam[14]; cam.cy = vw.cam[15];
            project_bwd_one(cam, mean, q4, scale, opac, vw.Wf, vw.Hf, eps2d, 2.0f * r0.z, r0.w, 2.0f * r1.x, r3.x,
                            g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, 0.0f, pg);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { pg.mean[c] = tail_group_sum<VPG>(pg.mean[c]); pg.scale[c] = tail_group_sum<VPG>(pg.scale[c]); }
#pragma unroll
    for (int c = 0; c < 4; ++c) pg.quat[c] = tail_group_sum<VPG>(pg.quat[c]);
    pg.op = tail_group_sum<VPG>(pg.op);
    if (n < N) {
        if (view == 0 % VPG) {                                      // the shading kernel wrote its part of v_means: always added
#pragma unroll
            for (int c = 0; c < 3; ++c) v_means[3 * (size_t)n + c] += pg.mean[c];
        }
        if (view == 1 % VPG) {
            float4 q = accumulate ? *reinterpret_cast<float4*>(v_quats + 4 * (size_t)n) : make_float4(0.f, 0.f, 0.f, 0.f);
            q.x += pg.quat[0]; q.y += pg.quat[1]; q.z += pg.quat[2]; q.w += pg.quat[3];
            *reinterpret_cast<float4*>(v_quats + 4 * (size_t)n) = q;
        }
        if (view == 2 % VPG) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v_scales[3 * (size_t)n + c] = (accumulate ? v_scales[3 * (size_t)n + c] : 0.0f) + pg.scale[c];
        }
        if (view == 3 % VPG) v_opacities[n] = (accumulate ? v_opacities[n] : 0.0f) + pg.op;
    }
}

template <int BLOCK, int VPG, bool DIFFUSE>
__global__ void __launch_bounds__(BLOCK)
tail_shade_pairs_kernel(int N, TailViewsDev views, const float* __restrict__ means, const float* __restrict__ normals,
                        const float* __restrict__ kd, const float* __restrict__ ks, float min_roughness, float max_metallic, EnvDev env,
                        int rec_stride, float* __restrict__ v_means, float* __restrict__ v_normals, float* __restrict__ v_kd,
                        float* __restrict__ v_ks, EnvGradDev eg, int accumulate, int mode_rt)
{
    const int mode = DIFFUSE ? GS_MODE_DIFFUSE : mode_rt;
    __builtin_assume(DIFFUSE || mode != GS_MODE_DIFFUSE);
    extern __shared__ __attribute__((aligned(16))) float s_grad[];
    __shared__ TailViewLds s_views[VPG];
    __shared__ TailLevelLds s_levels[GS_MAX_LEVELS];
    for (int i = threadIdx.x; i < eg.lds_floats; i += blockDim.x) s_grad[i] = 0.0f;
#pragma unroll
    for (int l = 0; l < GS_MAX_LEVELS; ++l) {                       // the level of a pair is per-lane data: a table, not kernel arguments
        if ((int)threadIdx.x == 64 + l) {
            TailLevelLds& d = s_levels[l];
            if (DIFFUSE) { d.tex = (unsigned long long)env.base; d.gtex = (unsigned long long)eg.base; d.priv_off = -1; d.R = env.base_res; d.lds_off = eg.lds_base; }
            else { d.tex = (unsigned long long)env.levels[l]; d.gtex = (unsigned long long)eg.levels[l]; d.priv_off = eg.priv ? eg.priv_level[l] : -1; d.R = env.res[l]; d.lds_off = eg.lds_level[l]; }
        }
    }
#pragma unroll
    for (int k = 0; k < VPG; ++k) {
        if ((int)threadIdx.x == k) {                                // k is a constant: the view's pointers come from scalar loads
            const TailViewDev& h = views.v[k];
            TailViewLds& d = s_views[k];
            d.vis = (unsigned long long)h.vis; d.v_packed = (unsigned long long)h.v_packed; d.packed_index = (unsigned long long)h.packed_index;
            const GsCam c = load_cam(h.viewmat, h.K);
#pragma unroll
            for (int i = 0; i < 9; ++i) d.cam[i] = c.R[i];
            d.cam[9] = c.t[0]; d.cam[10] = c.t[1]; d.cam[11] = c.t[2];
            d.cam[12] = c.fx; d.cam[13] = c.fy; d.cam[14] = c.cx; d.cam[15] = c.cy;
            d.cam_pos[0] = h.cam_pos[0]; d.cam_pos[1] = h.cam_pos[1]; d.cam_pos[2] = h.cam_pos[2];
            d.Wf = (float)h.W; d.Hf = (float)h.H; d.pad = 0.0f;
        }
    }
    __syncthreads();
    float* const stage = s_grad + eg.stage_off + (threadIdx.x >> 6) * 640;
    float* const priv = eg.priv ? eg.priv + (long long)gs_xcc_id() * eg.priv_stride : nullptr;
    constexpr int G = BLOCK / VPG;                                  // Gaussians per block and trip
    const int view = (int)threadIdx.x & (VPG - 1), gl = (int)threadIdx.x / VPG;
    const bool view_on = view < views.n;
    const TailViewLds& vw = s_views[view];
    const int stride = (int)gridDim.x * G;
    const int n_iter = (N + stride - 1) / stride;                   // wave-uniform: every lane reaches the wave-aggregated scatter
    for (int it = 0; it < n_iter; ++it) {
        const int n = it * stride + (int)blockIdx.x * G + gl;
        const bool live = n < N && view_on;
        const int slot = live ? reinterpret_cast<const int32_t*>(vw.packed_index)[n] : -1;
        float mean[3] = { 0, 0, 0 }, normal[3] = { 0, 0, 0 }, kdn[3] = { 0, 0, 0 }, ksn[2] = { 0, 0 };
        if (live) {
            mean[0] = means[3 * (size_t)n]; mean[1] = means[3 * (size_t)n + 1]; mean[2] = means[3 * (size_t)n + 2];
            normal[0] = normals[3 * (size_t)n]; normal[1] = normals[3 * (size_t)n + 1]; normal[2] = normals[3 * (size_t)n + 2];
            kdn[0] = kd[3 * (size_t)n]; kdn[1] = kd[3 * (size_t)n + 1]; kdn[2] = kd[3 * (size_t)n + 2];
            const float2 ks2 = *reinterpret_cast<const float2*>(ks + 2 * (size_t)n);
            ksn[0] = ks2.x; ksn[1] = ks2.y;
        }
        float a_mean[3] = { 0, 0, 0 }, a_n[3] = { 0, 0, 0 }, a_kd[3] = { 0, 0, 0 }, a_ks[2] = { 0, 0 };
        // ---- the pyramid levels one at a time: fetch + contract with the colour cotangent, then the texel scatter of that level by
        //      all lanes together
        const float* vp = reinterpret_cast<const float*>(vw.v_packed) + (size_t)(slot >= 0 ? slot : 0) * rec_stride;
        bool work = false;
        PairPre pre;
        float g[3] = { 0, 0, 0 };
        if (slot >= 0) {
            g[0] = vp[6]; g[1] = vp[7]; g[2] = vp[8];
            if (g[0] != 0.0f || g[1] != 0.0f || g[2] != 0.0f) {
                const float cp[3] = { vw.cam_pos[0], vw.cam_pos[1], vw.cam_pos[2] };
                shade_pair_pre(mean, normal, kdn, ksn, cp, min_roughness, max_metallic, mode, env, g, pre);
                work = true;
            }
        }
        float o_acc[3] = { 0, 0, 0 }, vd_acc[3] = { 0, 0, 0 }, v_mip = 0.0f;
#pragma unroll 1
        for (int lv = 0; lv < (DIFFUSE ? 1 : 2); ++lv) {
            const int l = work ? (lv == 0 ? pre.l0 : pre.l1) : -1;
            const bool on = l >= 0;
            const TailLevelLds& lt = s_levels[on ? l : 0];
            const float w = lv == 0 ? (pre.l1 < 0 ? 1.0f : 1.0f - pre.f) : pre.f;
            CubeFp fp;
            fp.valid = false;
            if (on) {
                float o[3], sv, dd[3];
                cube_fetch_vjp(reinterpret_cast<const float*>(lt.tex), lt.R, pre.dir, pre.v, o, sv, dd, fp);
#pragma unroll
                for (int c = 0; c < 3; ++c) { o_acc[c] = fmaf(w, o[c], o_acc[c]); vd_acc[c] = fmaf(w, dd[c], vd_acc[c]); }
                v_mip += lv == 0 ? -sv : sv;
            }
            const bool in_lds = on && lt.lds_off >= 0;
            if (in_lds) cube_scatter_lds(s_grad + lt.lds_off, fp, pre.v, w);
            const bool glob = on && !in_lds;
            const bool loc = glob && priv != nullptr && lt.priv_off >= 0;
            cube_scatter_wave_tagged(glob ? (loc ? priv + lt.priv_off : reinterpret_cast<float*>(lt.gtex)) : nullptr, loc, fp, pre.v, w, glob, stage);
        }
        if (work) shade_pair_post(pre, normal, kdn, g, mode, min_roughness, max_metallic, o_acc, vd_acc, v_mip, a_mean, a_n, a_kd, a_ks);
        // ---- sum over the views of the Gaussian (adjacent lanes), one lane of the group stores each array
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a_mean[c] = tail_group_sum<VPG>(a_mean[c]); a_n[c] = tail_group_sum<VPG>(a_n[c]); a_kd[c] = tail_group_sum<VPG>(a_kd[c]);
        }
        a_ks[0] = tail_group_sum<VPG>(a_ks[0]); a_ks[1] = tail_group_sum<VPG>(a_ks[1]);
        if (n < N) {
            if (view == 0 % VPG) {                                  // accumulate: read-modify-write by the owning lane
#pragma unroll
                for (int c = 0; c < 3; ++c) v_means[3 * (size_t)n + c] = (accumulate ? v_means[3 * (size_t)n + c] : 0.0f) + a_mean[c];
            }
            if (view == 1 % VPG) {
                float2 o = accumulate ? *reinterpret_cast<float2*>(v_ks + 2 * (size_t)n) : make_float2(0.f, 0.f);
                o.x += a_ks[0]; o.y += a_ks[1];
                *reinterpret_cast<float2*>(v_ks + 2 * (size_t)n) = o;
            }
            if (view == 2 % VPG) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v_normals[3 * (size_t)n + c] = (accumulate ? v_normals[3 * (size_t)n + c] : 0.0f) + a_n[c];
            }
            if (view == 3 % VPG) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v_kd[3 * (size_t)n + c] = (accumulate ? v_kd[3 * (size_t)n + c] : 0.0f) + a_kd[c];
            }
        }
    }
    // ---- flush the private copies
    __syncthreads();
    if (eg.lds_base >= 0) {
        const int cnt = 18 * env.base_res * env.base_res;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const float x = s_grad[eg.lds_base + i];
            if (x != 0.0f) gs_atomic_add(eg.base + i, x);
        }
    }
    for (int l = 0; l < env.L; ++l) {
        if (eg.lds_level[l] < 0) continue;
        const int cnt = 18 * env.res[l] * env.res[l];
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const float x = s_grad[eg.lds_level[l] + i];
            if (x != 0.0f) gs_atomic_add(eg.levels[l] + i, x);
        }
    }
}

extern "C" int gs_tail_bwd_multi_parts(int parts, int N, int n_views, const GsTailView* views, const float* means, const float* quats,
                                       const float* scales, const float* opacities, const float* normals, const float* kd, const float* ks,
                                       float min_roughness, float max_metallic, int mode, const GsEnv* env, float eps2d, int rec_stride,
                                       float* v_means, float* v_quats, float* v_scales, float* v_opacities, float* v_normals, float* v_kd,
                                       float* v_ks, int accumulate, const GsEnvGrad* env_grad, void* priv_ws, size_t priv_ws_bytes, void* stream)
{
    GS_CHECK_ARG(parts >= 1 && parts <= 7 && (parts & 3) != 0, "parts: bit 0 = shading backward, bit 1 = projection backward, bit 2 = background launch");
    GS_CHECK_ARG(N >= 0 && n_views >= 1 && mode >= 0 && mode <= 2 && rec_stride >= 12 && (rec_stride % 4) == 0, "bad sizes, mode or record stride");
    GS_CHECK_ARG(views != nullptr && env_grad != nullptr, "null argument");
    EnvDev e;
    GS_CHECK_ARG(env_to_dev(env, e) == 0, "bad GsEnv");
    EnvGradDev eg;
    eg.base = env_grad->base;
    for (int l = 0; l < GS_MAX_LEVELS; ++l) eg.levels[l] = l < e.L ? env_grad->levels[l] : nullptr;
    if (mode == GS_MODE_DIFFUSE) GS_CHECK_ARG(eg.base != nullptr, "env_grad->base required in diffuse mode");
    else for (int l = 0; l < e.L; ++l) GS_CHECK_ARG(eg.levels[l] != nullptr, "env_grad->levels[l] required");
    ShadeBwdPlan plan;
    { const int rc = shade_bwd_plan(e, mode, N, nullptr, 0, eg, plan, true, GS_TAILP_BLOCK); if (rc != GS_OK) return rc; }
    {
        const size_t floats = tail_priv_floats(e, mode, eg.priv_level);
        if (priv_ws != nullptr && floats > 0) {
            if (priv_ws_bytes < floats * sizeof(float) * GS_XCD_COPIES) { gs_set_error("gs_tail_bwd_multi: private workspace too small"); return GS_ENOSPC; }
            eg.priv = (float*)priv_ws; eg.priv_stride = (long long)floats;
        } else {
            eg.priv = nullptr; eg.priv_stride = 0;
            for (int l = 0; l < GS_MAX_LEVELS; ++l) eg.priv_level[l] = -1;
        }
    }
    if (N == 0) return GS_OK;
    hipStream_t s = (hipStream_t)stream;
    for (int v0 = 0; v0 < n_views; v0 += GS_TAIL_MAX_VIEWS) {             // more than 8 views: further launches ADD to the first
        TailViewsDev tv;
        tv.n = (n_views - v0) < GS_TAIL_MAX_VIEWS ? (n_views - v0) : GS_TAIL_MAX_VIEWS;
        for (int k = 0; k < tv.n; ++k) {
            const GsTailView& h = views[v0 + k];
            GS_CHECK_ARG(h.viewmat && h.K && h.cam_pos && h.vis_records && h.v_packed && h.packed_index && h.W > 0 && h.H > 0, "bad GsTailView");
            tv.v[k] = TailViewDev{ h.viewmat, h.K, h.cam_pos, (const float4*)h.vis_records, h.v_packed, h.packed_index, h.W, h.H };
        }
        for (int k = tv.n; k < GS_TAIL_MAX_VIEWS; ++k) tv.v[k] = tv.v[0];
        const int acc = (accumulate || v0 > 0) ? 1 : 0;
#define GS_TAILP_LAUNCH(B, VPG, DIFF)                                                                                                    \
        do {                                                                                                                            \
            const int groups = gs_cdiv(N, B / VPG);                                                                                     \
            int max_blocks = eg.lds_floats > 0 ? 256 * (int)fmax(1.0, floor(160.0 * 1024.0 / (double)(plan.lds_bytes + 2048))) : 2048; \
            if ((parts & 4) && eg.lds_floats > 0) max_blocks = tail_background_blocks();                                                \
            if (parts & 1) {                                                                                                            \
                GS_CHECK_HIP(hipFuncSetAttribute((const void*)tail_shade_pairs_kernel<B, VPG, DIFF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes)); \
                hipLaunchKernelGGL((tail_shade_pairs_kernel<B, VPG, DIFF>), dim3(groups < max_blocks ? groups : max_blocks), dim3(B), plan.lds_bytes, s, N, tv, \
                                   means, normals, kd, ks, min_roughness, max_metallic, e, rec_stride, v_means, v_normals, v_kd, v_ks, eg, acc, mode); \
                GS_CHECK_LAUNCH();                                                                                                      \
            }                                                                                                                           \
            if (parts & 2) {                                                                                                            \
                hipLaunchKernelGGL((tail_proj_pairs_kernel<256, VPG>), dim3(gs_cdiv(N, 256 / VPG)), dim3(256), 0, s, N, tv, means, quats, scales, \
                                   opacities, eps2d, rec_stride, v_means, v_quats, v_scales, v_opacities, acc);                                \
                GS_CHECK_LAUNCH();                                                                                                      \
            }                                                                                                                           \
        } while (0)
#define GS_TAILP_MODE(VPG) do { if (mode == GS_MODE_DIFFUSE) GS_TAILP_LAUNCH(GS_TAILP_BLOCK, VPG, true); else GS_TAILP_LAUNCH(GS_TAILP_BLOCK, VPG, false); } while (0)
        if (tv.n <= 1) GS_TAILP_MODE(1); else if (tv.n <= 2) GS_TAILP_MODE(2); else if (tv.n <= 4) GS_TAILP_MODE(4); else GS_TAILP_MODE(8);
#undef GS_TAILP_MODE
#undef GS_TAILP_LAUNCH
    }
    return GS_OK;
}

extern "C" int gs_tail_bwd_multi(int N, int n_views, const GsTailView* views, const float* means, const float* quats, const float* scales,
                                 const float* opacities, const float* normals, const float* kd, const float* ks, float min_roughness,
                                 float max_metallic, int mode, const GsEnv* env, float eps2d, int rec_stride, float* v_means, float* v_quats,
                                 float* v_scales, float* v_opacities, float* v_normals, float* v_kd, float* v_ks, int accumulate,
                                 const GsEnvGrad* env_grad, void* priv_ws, size_t priv_ws_bytes, void* stream)
{
    return gs_tail_bwd_multi_parts(3, N, n_views, views, means, quats, scales, opacities, normals, kd, ks, min_roughness, max_metallic, mode, env,
                                   eps2d, rec_stride, v_means, v_quats, v_scales, v_opacities, v_normals, v_kd, v_ks, accumulate, env_grad,
                                   priv_ws, priv_ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Chain rule of the once-per-step activations (rfstudio/model/gsplat.py:336-339: scales.exp(), sigmoid(opacities)) in ONE launch:
//   v_scales = v_scales_act * scales_act,   v_opacities = (v_opac_act * opac_act) * (1 - opac_act)
// -- the products torch formed with five elementwise launches at the end of every step, in the same order (contraction off).
__global__ void __launch_bounds__(256)
activation_chain_kernel(int64_t n3, int64_t n, const float* __restrict__ g_scales_act, const float* __restrict__ scales_act,
                        const float* __restrict__ g_opac_act, const float* __restrict__ opac_act, float* __restrict__ v_scales,
                        float* __restrict__ v_opac)
{
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) v_scales[i] = g_scales_act[i] * scales_act[i];
    if (i < n) { const float o = opac_act[i]; v_opac[i] = (g_opac_act[i] * o) * (1.0f - o); }
}

extern "C" int gs_activation_chain(int64_t N, const float* g_scales_act, const float* scales_act, const float* g_opac_act,
                                   const float* opac_act, float* v_scales, float* v_opacities, void* stream)
{
    GS_CHECK_ARG(N >= 0, "bad N");
    if (N == 0) return GS_OK;
    GS_CHECK_ARG(g_scales_act && scales_act && g_opac_act && opac_act && v_scales && v_opacities, "null argument");
    hipLaunchKernelGGL(activation_chain_kernel, dim3(gs_cdiv(3 * N, 256)), dim3(256), 0, (hipStream_t)stream, 3 * N, N, g_scales_act,
                       scales_act, g_opac_act, opac_act, v_scales, v_opacities);
    GS_CHECK_LAUNCH();
    return GS_OK;
}
