/*
 * geosplat_hip.h -- C-ABI of libgeosplat_hip.so: the MI355X (gfx950) implementation of the
 * GeoSplatting render-and-backward hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. the PyTorch-ROCm allocator) unless the
 *     parameter is documented "host"; all tensors are contiguous fp32 / int32 / int64 as noted;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream);
 *   - no allocation, no host synchronisation and no exceptions inside; every call returns 0 on success or
 *     a negative GS_E* code, with a message available from gs_last_error() (thread-local);
 *   - data-dependent sizes (V visible Gaussians, I tile intersections) are produced ON THE DEVICE in
 *     `counts[2]`; the caller reads them (one 16-byte copy) before sizing the I-length buffers, exactly
 *     where the reference's upstream performs its own host sync.
 *
 * Reference interfaces replaced (file:line relative to /root/reference):
 *   gs_project_fwd / gs_isect_emit / gs_isect_sort / gs_isect_offsets / gs_raster_fwd
 *        = the stages executed by `gsplat.rasterization(...)` as called at
 *          rfstudio/model/gsplat.py:334-355 (also :240-261, :151-172; rfstudio/model/geosplat.py:276-295)
 *   gs_raster_bwd / gs_project_bwd
 *        = the autograd backward of that call, reached from rfstudio/optim/optimizer.py:107
 *   gs_shade_fwd / gs_shade_bwd
 *        = RenderableAttrs.splat arithmetic rfstudio/model/geosplat.py:80-122 including the two
 *          `dr.texture` calls (:93-98 and rfstudio/graphics/_mesh/_texture.py:596-611)
 *   gs_tonemap_fwd / gs_tonemap_bwd
 *        = _tone_mapping_naive / _tone_mapping_aces rfstudio/model/geosplat.py:474-480
 *   gs_cubemap_mip_fwd, gs_cube_sample_linear, gs_diffuse_cubemap_fwd/_bwd, gs_specular_bounds,
 *   gs_specular_cubemap_fwd/_bwd
 *        = the pybind entry points diffuse_cubemap_fwd/bwd, specular_bounds, specular_cubemap_fwd/bwd
 *          (rfstudio/graphics/_mesh/_splitsum/c_src/torch_bindings.cpp:265-271) and _CubeMapMip
 *          (rfstudio/graphics/_mesh/_texture.py:199-226)
 */
#ifndef GEOSPLAT_HIP_H
#define GEOSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_OK            0
#define GS_EINVAL       -1   /* bad argument */
#define GS_ENOSPC       -2   /* workspace too small */
#define GS_ELAUNCH      -3   /* HIP launch / runtime error */

#define GS_MAX_LEVELS   16   /* maximum mip levels of the split-sum pyramid */
#define GS_MAX_CHANNELS 32   /* maximum colour channels D of the compositor */

/* shading modes of RenderableAttrs.splat (rfstudio/model/geosplat.py:62) */
#define GS_MODE_PBR      0
#define GS_MODE_DIFFUSE  1
#define GS_MODE_SPECULAR 2
/* tone mapping (rfstudio/model/geosplat.py:63) */
#define GS_TONE_NONE     0
#define GS_TONE_NAIVE    1
#define GS_TONE_ACES     2

const char* gs_last_error(void);
int         gs_version(void);

/* ------------------------------------------------------------------ A1 + A1' + A2(count) ---------- */
/* Bytes of scratch gs_project_fwd needs for its single-pass chained scan (zeroed by the call itself). */
size_t gs_project_ws_bytes(int N);

/* Fused projection + cull + anti-alias compensation + packed compaction (ascending Gaussian index)
 * + opacity*compensation + colour gather + tiles-per-Gaussian + inclusive tile cumsum.
 * Packed outputs need capacity N.  counts[0] = V, counts[1] = I (int64).
 * colors / colors_packed may both be NULL (D ignored).  packed_index[N] (nullable) receives the packed
 * slot of every Gaussian or -1.  viewmat: row-major 4x4 world->camera (OpenCV), K: row-major 3x3. */
int gs_project_fwd(int N, const float* means, const float* quats, const float* scales, const float* opacities,
                   const float* colors, int D, const float* viewmat, const float* K, int W, int H, int tile_size,
                   float eps2d, float near_plane, float far_plane, float radius_clip,
                   int32_t* gaussian_ids, int32_t* radii, float* means2d, float* depths, float* conics,
                   float* compensations, float* opacities_packed, float* colors_packed,
                   int32_t* tiles_per_gauss, int64_t* cum_tiles, int32_t* packed_index,
                   void* ws, size_t ws_bytes, int64_t* counts, void* stream);

/* The same, additionally writing the compositor's 64-byte per-visible records {mx,my,a/2,b | c/2,opacity,hx,hy | c0,c1,c2,- | pad}
 * (vis_records [N,16] floats, nullable) straight from the projection's registers; gs_raster_prepare_vis then skips its own packing
 * pass.  Colours travel in the record for D <= 3. */
int gs_project_fwd_vis(int N, const float* means, const float* quats, const float* scales, const float* opacities,
                       const float* colors, int D, const float* viewmat, const float* K, int W, int H, int tile_size,
                       float eps2d, float near_plane, float far_plane, float radius_clip,
                       int32_t* gaussian_ids, int32_t* radii, float* means2d, float* depths, float* conics,
                       float* compensations, float* opacities_packed, float* colors_packed,
                       int32_t* tiles_per_gauss, int64_t* cum_tiles, int32_t* packed_index, float* vis_records,
                       void* ws, size_t ws_bytes, int64_t* counts, void* stream);

/* ------------------------------------------------------------------ A2 (emit) ---------------------- */
/* isect_ids[i] = (tile_id << 32) | float_bits(depth), flatten_ids[i] = packed index; emission order =
 * ascending packed index, tiles row-major. */
int gs_isect_emit(int V, const float* means2d, const int32_t* radii, const float* depths,
                  const int64_t* cum_tiles, int tile_size, int tile_w, int tile_h,
                  int64_t* isect_ids, int32_t* flatten_ids, void* stream);

/* ------------------------------------------------------------------ A2 + A3 (depth-major binning) -- */
/* The sorted intersection list straight from the projection outputs, WITHOUT emitting the unsorted keys: the Gaussians are
 * ordered by depth once (V-sized), their intersections are emitted in that order as (tile, index) items and two stable
 * passes over the tile bits finish the (tile | depth | packed index) order -- bit-identical to gs_isect_emit + gs_isect_sort
 * (csrc/gs_sort.hip).  tiles_per_gauss [V] and n_isects (= their sum) come from gs_project_fwd. */
size_t gs_isect_bin_ws_bytes(int V, int64_t n_isects, int tile_w, int tile_h);
int gs_isect_bin(int V, const float* means2d, const int32_t* radii, const float* depths, const int32_t* tiles_per_gauss,
                 int64_t n_isects, int tile_size, int tile_w, int tile_h, int64_t* isect_ids_sorted,
                 int32_t* flatten_ids_sorted, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ A3 ----------------------------- */
size_t gs_sort_ws_bytes(int64_t n_isects, int tile_w, int tile_h);
/* Stable ascending sort of (isect_ids, flatten_ids) over the 32 + tile_bits significant key bits. */
int gs_isect_sort(int64_t n_isects, const int64_t* isect_ids, const int32_t* flatten_ids,
                  int64_t* isect_ids_sorted, int32_t* flatten_ids_sorted, int tile_w, int tile_h,
                  void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ A4 ----------------------------- */
/* offsets[t] = first sorted position whose tile id >= t  (t in [0, n_tiles)). */
int gs_isect_offsets(int64_t n_isects, const int64_t* isect_ids_sorted, int n_tiles, int32_t* offsets,
                     void* stream);

/* ------------------------------------------------------------------ A5 / A6 ------------------------ */
/* Bytes of the compositor workspace: the per-intersection record stream in sorted order (48 B each) plus the
 * longest-first tile order.  gs_raster_fwd WRITES it, gs_raster_bwd READS it (keep it alive in between). */
size_t gs_raster_ws_bytes(int64_t n_isects, int V, int W, int H, int tile_size);

/* Front-to-back alpha compositing of the per-tile sorted lists.  colors: packed [V,D]; opacities: packed
 * (already multiplied by the compensation); background: nullable [D].
 * render [H,W,D], alphas [H,W], last_ids [H,W] (index into the sorted list of the last composited entry). */
int gs_raster_fwd(int W, int H, int tile_size, int D, int V, const float* means2d, const float* conics,
                  const float* opacities, const float* colors, const float* background,
                  int64_t n_isects, const int32_t* offsets, const int32_t* flatten_ids,
                  float* render, float* alphas, int32_t* last_ids, void* ws, size_t ws_bytes, void* stream);

/* The two halves of gs_raster_fwd for a caller that spreads a view over several streams: gs_raster_prepare fills
 * the workspace (per-visible records, sorted record stream, tile order: HBM-bound), gs_raster_composite runs the
 * compositor on it (VALU-bound).  gs_raster_fwd == prepare followed by composite on one stream. */
int gs_raster_prepare(int W, int H, int tile_size, int D, int V, const float* means2d, const float* conics,
                      const float* opacities, const float* colors, int64_t n_isects, const int32_t* offsets,
                      const int32_t* flatten_ids, void* ws, size_t ws_bytes, void* stream);
int gs_raster_prepare_vis(int W, int H, int tile_size, int D, int V, const float* vis_records /* gs_project_fwd_vis */,
                          int64_t n_isects, const int32_t* offsets, const int32_t* flatten_ids, void* ws, size_t ws_bytes,
                          void* stream);
int gs_raster_composite(int W, int H, int tile_size, int D, int V, const float* colors, const float* background,
                        int64_t n_isects, const int32_t* offsets, float* render, float* alphas, int32_t* last_ids,
                        const void* ws, size_t ws_bytes, void* stream);

/* Floats per packed per-Gaussian gradient record written by gs_raster_bwd: (6 + D) rounded up to 16. */
int gs_raster_grad_stride(int D);

/* Stored-state backward (uses alphas + last_ids + the workspace of the forward).  The per-visible-Gaussian
 * gradients come out as ONE packed record per Gaussian (one or two cache lines, so the 6+D fp32 atomics of a
 * survivor merge into 1-2 memory-side requests):
 *     v_packed[g*stride + 0..1] = v_means2d   [2..4] = v_conics   [5] = v_opacities   [6..6+D) = v_colors
 * with stride = gs_raster_grad_stride(D); the buffer [V*stride] is zeroed by the call.  colors is only read when
 * D > 3 (for D <= 3 the colours travel inside the record stream). */
int gs_raster_bwd(int W, int H, int tile_size, int D, int V, const float* colors, const float* background,
                  int64_t n_isects, const int32_t* offsets, const float* alphas, const int32_t* last_ids,
                  const float* v_render, const float* v_alphas, float* v_packed, const void* ws, size_t ws_bytes,
                  void* stream);

/* The compositor backward alone: ADDS into a v_packed the caller has zeroed (or that already holds another view's
 * packed gradients of the same visible set); same arguments. */
int gs_raster_bwd_acc(int W, int H, int tile_size, int D, int V, const float* colors, const float* background,
                      int64_t n_isects, const int32_t* offsets, const float* alphas, const int32_t* last_ids,
                      const float* v_render, const float* v_alphas, float* v_packed, const void* ws, size_t ws_bytes,
                      void* stream);

/* ------------------------------------------------------------------ capacity protocol --- */
/* The data-dependent sizes of a view -- V visible Gaussians, I tile intersections -- exist only on the device after
 * gs_project_fwd (counts = {V, I}, int64).  The entry points above take them as host scalars (gsplat's call shape: one
 * read-back per view).  The *_cap variants below take CAPACITIES instead: V_cap (N always suffices) and n_isects_cap size
 * the buffers, the workspaces and the launch grids; every kernel reads the actual count from counts_dev and clamps it to
 * the capacity.  No host synchronisation, fixed launch shapes (capturable in a hipGraph).  If a view needs more than
 * n_isects_cap, gs_isect_bin_cap sets status_dev = {GS_ENOSPC, max required n_isects, max required V} (int64[3], zeroed once by
 * the caller, sticky) and that view is composited from the first n_isects_cap intersections in emission order: memory-safe,
 * wrong image -- the caller polls status_dev at a convenient point (the engine: start of the next step), grows the capacity
 * and repeats the step.  Workspace sizes: the *_ws_bytes functions with the capacities. */
int gs_isect_bin_cap(int V_cap, const float* means2d, const int32_t* radii, const float* depths, const int64_t* counts_dev,
                     int64_t n_isects_cap, int tile_size, int tile_w, int tile_h, int64_t* isect_ids_sorted,
                     int32_t* flatten_ids_sorted, void* ws, size_t ws_bytes, int64_t* status_dev, void* stream);
int gs_isect_offsets_cap(int64_t n_isects_cap, const int64_t* counts_dev, const int64_t* isect_ids_sorted, int n_tiles,
                         int32_t* offsets, void* stream);
/* The pair for a caller that only composites (the step engine never returns `meta`): the binning without the 64-bit isect_ids -- the
 * last pass writes int32 tile ids instead (8 bytes per intersection instead of 12, no depth gather) -- and the tile offsets from
 * those.  flatten_ids_sorted / offsets are bit-identical to gs_isect_bin_cap + gs_isect_offsets_cap. */
int gs_isect_bin_tiles_cap(int V_cap, const float* means2d, const int32_t* radii, const float* depths, const int64_t* counts_dev,
                           int64_t n_isects_cap, int tile_size, int tile_w, int tile_h, int32_t* tile_ids_sorted,
                           int32_t* flatten_ids_sorted, void* ws, size_t ws_bytes, int64_t* status_dev, void* stream);
int gs_isect_offsets_tiles_cap(int64_t n_isects_cap, const int64_t* counts_dev, const int32_t* tile_ids_sorted, int n_tiles,
                               int32_t* offsets, void* stream);
int gs_raster_prepare_vis_cap(int W, int H, int tile_size, int D, int V_cap, const float* vis_records, int64_t n_isects_cap,
                              const int64_t* counts_dev, const int32_t* offsets, const int32_t* flatten_ids, void* ws,
                              size_t ws_bytes, void* stream);
int gs_raster_composite_cap(int W, int H, int tile_size, int D, int V_cap, const float* colors, const float* background,
                            int64_t n_isects_cap, const int64_t* counts_dev, const int32_t* offsets, float* render,
                            float* alphas, int32_t* last_ids, const void* ws, size_t ws_bytes, void* stream);
int gs_raster_bwd_cap(int W, int H, int tile_size, int D, int V_cap, const float* colors, const float* background,
                      int64_t n_isects_cap, const int64_t* counts_dev, const int32_t* offsets, const float* alphas,
                      const int32_t* last_ids, const float* v_render, const float* v_alphas, float* v_packed, const void* ws,
                      size_t ws_bytes, void* stream);
/* gs_raster_bwd_acc with the counts on the device: ADDS into a v_packed [V_cap, stride] the caller has zeroed (the engine
 * zeroes it on its front stream, under the compositor of the previous view). */
int gs_raster_bwd_acc_cap(int W, int H, int tile_size, int D, int V_cap, const float* colors, const float* background,
                          int64_t n_isects_cap, const int64_t* counts_dev, const int32_t* offsets, const float* alphas,
                          const int32_t* last_ids, const float* v_render, const float* v_alphas, float* v_packed, const void* ws,
                          size_t ws_bytes, void* stream);
/* The compositor with S4 (tone mapping, rfstudio/model/geosplat.py:123-133) inside, D = 3 and no background -- what
 * RenderableAttrs.splat runs between `rasterization` and the loss (rfstudio/model/geosplat.py:108-133): the forward also writes
 * image[P,4] = tonemap(render * exposure | alpha) (= gs_tonemap_fwd3 on its outputs, bit-identical), the backward takes the image
 * cotangent v_image[P,4] instead of v_render / v_alphas (= gs_tonemap_bwd3 + gs_raster_bwd_acc) and ADDS the exposure gradient to
 * the device scalar v_exposure.  Two dependent launches fewer per view on the step's critical stream.  counts_dev == NULL: V and
 * n_isects are exact; otherwise they are capacities and {V, I} are read on the device (capacity protocol). */
int gs_raster_composite_tone(int W, int H, int tile_size, int V, const float* colors, int64_t n_isects,
                             const int64_t* counts_dev, const int32_t* offsets, float* render, float* alphas,
                             int32_t* last_ids, int tone_mode, const float* exposure /*device scalar*/, float* image,
                             const void* ws, size_t ws_bytes, void* stream);
int gs_raster_bwd_tone_acc(int W, int H, int tile_size, int V, const float* colors, int64_t n_isects,
                           const int64_t* counts_dev, const int32_t* offsets, const float* render, const float* alphas,
                           const int32_t* last_ids, int tone_mode, const float* exposure, const float* v_image,
                           float* v_packed, float* v_exposure, const void* ws, size_t ws_bytes, void* stream);
/* The same pair with a CULL LOG between them: the forward appends {stream index, pixel mask} of every record that entered one of its
 * dense batches (per tile quadrant, in stream order) to log_ws (gs_raster_log_ws_bytes; written by the forward, read by the backward of
 * the SAME view), and the backward walks that log from its end instead of repeating the forward's cull and ellipse masks.  Same image
 * bit for bit, same gradients up to the order of the float atomics.  The forward also files every tile, by its longest quadrant
 * log, in bucket lists inside `ws` -- the order in which the backward launches its tiles; the lists are cleared by gs_raster_prepare*,
 * so they are used when exactly ONE forward ran on the prepared workspace (otherwise the backward falls back to the forward's order). */
size_t gs_raster_log_ws_bytes(int64_t n_isects, int W, int H, int tile_size);
int gs_raster_composite_tone_log(int W, int H, int tile_size, int V, const float* colors, int64_t n_isects,
                                 const int64_t* counts_dev, const int32_t* offsets, float* render, float* alphas,
                                 int32_t* last_ids, int tone_mode, const float* exposure, float* image, const void* ws,
                                 size_t ws_bytes, void* log_ws, size_t log_bytes, void* stream);
int gs_raster_bwd_tone_log_acc(int W, int H, int tile_size, int V, const float* colors, int64_t n_isects,
                               const int64_t* counts_dev, const int32_t* offsets, const float* render, const float* alphas,
                               const int32_t* last_ids, int tone_mode, const float* exposure, const float* v_image,
                               float* v_packed, float* v_exposure, const void* ws, size_t ws_bytes, const void* log_ws,
                               size_t log_bytes, void* stream);
int gs_project_bwd_cap(int N, const int64_t* counts_dev, int D, const float* means, const float* quats, const float* scales,
                       const float* opacities, const float* viewmat, const float* K, int W, int H, float eps2d,
                       const int32_t* gaussian_ids, const float* conics, const float* compensations, const float* v_packed,
                       int rec_stride, const float* v_depths, float* v_means, float* v_quats, float* v_scales,
                       float* v_opacities, float* v_colors, int accumulate, void* stream);

/* Self-test of the compositor's reciprocal (hardware v_rcp_f32 + Newton + Markstein correction, which replaces the IEEE
 * division 1 / (1 - alpha) of gsplat's rasterize_to_pixels backward): *mismatches_dev (device uint64) = number of floats
 * with bit patterns in [lo_bits, hi_bits] whose result differs from the correctly rounded quotient (counted twice, once
 * per packed component).  Test hook; no reference counterpart. */
int gs_selftest_rcp(uint32_t lo_bits, uint32_t hi_bits, uint64_t* mismatches_dev, void* stream);

/* Self-test of the compositor's canonical exponential exp(-sigma) (one spelled-out operation order instead of gsplat's
 * hardware `__expf`, so that the CPU oracle reproduces it bit for bit): over every float with bit pattern in
 * [lo_bits, hi_bits], out_dev[0] (device uint64) = sum_i bits(result_i) * (2 i + 1) mod 2^64 -- an order-independent checksum the
 * oracle computes from its own copy -- and out_dev[1] = the bit pattern of the largest relative error (a double) against the
 * float64 exponential.  Test hook; no reference counterpart. */
int gs_selftest_exp(uint32_t lo_bits, uint32_t hi_bits, uint64_t* out_dev, void* stream);

/* Self-test of the cube-map edge table (csrc/gs_cube.h): the texel across a face edge is looked up in a 24-entry integer table
 * instead of being re-projected in floating point as oracle/gs_oracle_shade.c defines it; *mismatches_dev (device uint64) = number
 * of (face, edge, position) triples of an R x R face, R <= 2048, for which the two disagree.  Test hook; no reference counterpart. */
int gs_selftest_cube_edges(int R, uint64_t* mismatches_dev, void* stream);

/* ------------------------------------------------------------------ A7 ----------------------------- */
/* Projection backward + gather backward from the packed records of gs_raster_bwd; dense outputs [N,*] are fully
 * written (zeros for culled Gaussians) -- no caller-side zeroing needed -- or, with accumulate != 0, ADDED to
 * v_means / v_quats / v_scales / v_opacities (several views per step share one gradient buffer; v_colors is always
 * written).  v_depths nullable. */
int gs_project_bwd(int N, int V, int D, const float* means, const float* quats, const float* scales,
                   const float* opacities, const float* viewmat, const float* K, int W, int H, float eps2d,
                   const int32_t* gaussian_ids, const float* conics, const float* compensations,
                   const float* v_packed, int rec_stride /* 0 = gs_raster_grad_stride(D) */, const float* v_depths,
                   float* v_means, float* v_quats, float* v_scales, float* v_opacities, float* v_colors,
                   int accumulate, void* stream);

/* ------------------------------------------------------------------ S1..S3 ------------------------- */
/* Split-sum pyramid description (host struct, device pointers inside). */
typedef struct GsEnv {
    const float* lut;                     /* FG LUT [lut_res, lut_res, 2], row = roughness, col = N.V */
    int          lut_res;
    const float* base;                    /* diffuse irradiance [6, base_res, base_res, 3] */
    int          base_res;
    int          num_levels;              /* L */
    const float* levels[GS_MAX_LEVELS];   /* specular level l: [6, res[l], res[l], 3] */
    int          res[GS_MAX_LEVELS];
    float        min_roughness;           /* TextureSplitSum.min_roughness (0.08) */
    float        max_roughness;           /* TextureSplitSum.max_roughness (0.5)  */
} GsEnv;

typedef struct GsEnvGrad {
    float* base;                          /* [6, base_res, base_res, 3], accumulated into (caller zeroes) */
    float* levels[GS_MAX_LEVELS];
} GsEnvGrad;

/* colors[N,3] = shade(means, normals, kd, ks | cam_pos[3] (device), env). */
int gs_shade_fwd(int N, const float* means, const float* normals, const float* kd, const float* ks,
                 const float* cam_pos, float min_roughness, float max_metallic, int mode,
                 const GsEnv* env /*host*/, float* colors, void* stream);

/* Bytes of the optional workspace of gs_shade_bwd: 8 XCD-private copies of the texel-gradient levels that do
 * not fit LDS (XCD-local atomics stay in L2; device-scope ones are 32-byte write-throughs to the fabric). */
size_t gs_shade_bwd_ws_bytes(const GsEnv* env /*host*/, int mode);

/* Recomputes the forward and chains v_colors[N,3].  v_means/v_normals/v_kd/v_ks are fully written (or added
 * to when accumulate != 0); texel gradients are always ACCUMULATED into env_grad.  ws may be NULL (plain
 * device-scope atomics). */
int gs_shade_bwd(int N, const float* means, const float* normals, const float* kd, const float* ks,
                 const float* cam_pos, float min_roughness, float max_metallic, int mode,
                 const GsEnv* env /*host*/, const float* v_colors,
                 float* v_means, float* v_normals, float* v_kd, float* v_ks,
                 const GsEnvGrad* env_grad /*host*/, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ S4 ----------------------------- */
/* out[P,4] = tonemap(rgba[P,4] * exposure) ; exposure is a DEVICE scalar. */
int gs_tonemap_fwd(int64_t P, int mode, const float* rgba, const float* exposure, float* out, void* stream);
/* The same on the rasterizer's own layout (render [P,3] + alphas [P] -> image [P,4]; backward -> v_render [P,3], v_alphas [P]):
 * no rgba concatenation / strided copies between the compositor and the tone map. */
int gs_tonemap_fwd3(int64_t P, int mode, const float* render, const float* alphas, const float* exposure, float* out, void* stream);
int gs_tonemap_bwd3(int64_t P, int mode, const float* render, const float* alphas, const float* exposure, const float* v_out,
                    float* v_render, float* v_alphas, float* v_exposure, int accumulate, void* stream);
/* v_exposure: device scalar, zeroed by the call (accumulate == 0) then accumulated into. */
int gs_tonemap_bwd(int64_t P, int mode, const float* rgba, const float* exposure, const float* v_out,
                   float* v_rgba, float* v_exposure, int accumulate, void* stream);

/* ------------------------------------------------------------------ S5 ----------------------------- */
int gs_cubemap_mip_fwd(int R, int C, const float* in, float* out, void* stream);
/* n_levels successive gs_cubemap_mip_fwd calls (C = 3) in one launch: outs[k] = [6, R >> (k+1), R >> (k+1), 3], k < n_levels <= 5,
 * R a multiple of 2^n_levels (`outs` is a HOST array of device pointers).  Bit-identical to the chain of single calls
 * (the `while mips[-1].resolution > min_resolution` loop of TextureCubeMap.as_splitsum, rfstudio/graphics/_mesh/_texture.py:536-541). */
int gs_cubemap_mip_chain_fwd(int R, int n_levels, const float* in, float* const* outs, void* stream);
/* out[n,3] = seam-aware bilinear cube lookup of tex[6,R,R,3] at dirs[n,3] (used by the mip backward). */
int gs_cube_sample_linear(int64_t n, const float* tex, int R, const float* dirs, float scale, float* out,
                          void* stream);
/* _CubeMapMip.backward in one call: v_in[6,2R,2R,3] = 0.25 * cube-sample(v_out[6,R,R,3]) at fine texel dirs,
 * ADDED to v_in when accumulate != 0. */
int gs_cubemap_mip_bwd(int R, const float* v_out, float* v_in, int accumulate, void* stream);
int gs_diffuse_cubemap_fwd(int R, const float* cubemap, float* out, void* stream);
int gs_diffuse_cubemap_bwd(int R, const float* v_out, float* v_cubemap, int accumulate, void* stream);
/* bounds[6,R,R,24] (float-encoded ints, layout of the reference: per texel and face xmin, xmax, ymin, ymax of the lobe's texels).
 * gs_specular_bounds is shaped like SpecularBoundsKernel (cubemap.cu:181-244: every texel re-derives the corner directions of every
 * 16x16 tile); gs_specular_bounds_fast gives bit-identical boxes from per-tile / per-4x4-tile-group corner boxes computed once
 * (ws: gs_specular_bounds_ws_bytes(R) bytes of device scratch) and the cached direction table -- ~100x faster, the one the host uses. */
int gs_specular_bounds(int R, float costheta_cutoff, float* bounds, void* stream);
size_t gs_specular_bounds_ws_bytes(int R);
int gs_specular_bounds_fast(int R, float costheta_cutoff, const float* dir_table, float* bounds, void* ws, size_t ws_bytes, void* stream);
/* table[6,R,R,4] = {unit direction xyz, pixel_area}: depends on R only -- compute once, reuse every step. */
int gs_cube_dir_table(int R, float* table, void* stream);
/* out[6,R,R,4] = (sum rgb*w, sum w) */
int gs_specular_cubemap_fwd(int R, const float* cubemap, const float* bounds, const float* dir_table,
                            float roughness, float costheta_cutoff, float* out, void* stream);
/* v_out_rgb[6,R,R,3] = gradient w.r.t. the un-normalised rgb sums; v_cubemap written (or accumulated).  Atomic-free gather:
 * a source texel visits the outputs inside its own box grown by a margin (every texel for R < 64) and keeps those whose box
 * contains it -- the exact adjoint of gs_specular_cubemap_fwd (the reference scatters with atomicAdd, cubemap.cu:300-350). */
int gs_specular_cubemap_bwd(int R, const float* bounds, const float* dir_table, const float* v_out_rgb,
                            float roughness, float costheta_cutoff, float* v_cubemap, int accumulate, void* stream);

/* Tiled pair-weight tables of the specular prefilter (csrc/gs_splitsum_tiles.hip; replaces calling SpecularCubemapFwd/BwdKernel,
 * rfstudio/graphics/_mesh/_splitsum/c_src/cubemap.cu:246-350, with weights recomputed every step).  The pair weights
 * w(o,i) = g(o,i) * pixel_area(i) / 4 depend on (R, roughness, cutoff) only.  Units: a TILE = nb BLOCKS of 8x8 output texels of one
 * face in rows of bw blocks (tiles[n][4] = {face, x0, y0, 0}; block b covers x0 + 8 (b % bw) .., y0 + 8 (b / bw) ..); per
 * (tile, source face, block) -- "slot" (tile * 6 + face) * nb + b -- a list of ROWS, each one weight per lane (= output texel of
 * the block, lane = (y & 7) * 8 + (x & 7)) for the source texel anchor_lane + (dx, dy), anchor = min corner of the lane's lobe box
 * on that face (gs_specular_bounds), descriptor = (dy * pitch + dx) * 16 = the row's byte offset in the staged source rectangle.
 * Row lists are padded to multiples of 8 rows (zero weights); weights are stored in row pairs, [row / 2][lane][2].
 *   gs_specular_tiles_count : row_counts[n_tiles*6*nb] (un-padded), extents[n_tiles*6*nb][4] = {xmin, xmax, ymin, ymax} of the source
 *                             texels a slot's rows can address, *pairs += number of (output, source) pairs.  backward != 0: the
 *                             TRANSPOSED operator as a gather -- lane = source texel, partners = outputs inside the lane's own box
 *                             grown by `margin` texels (margin >= R: every texel of every face) whose box contains the lane's texel:
 *                             the exact adjoint of the forward.
 *   gs_specular_tiles_fill  : desc[rows], weights[rows * 64] (caller zero-fills both; row_begin[slot] = exclusive sum of the padded
 *                             counts; segments as below): g(o,i), the mirror-invariant factor of the weight.
 *   gs_specular_tiles_check : out2[0] = texels whose direction-table entry is not the exact sign-flipped image under the cube's
 *                             seven reflections, out2[1] = lobe boxes that are not the reflected box of the mirrored texel.  Both 0
 *                             <=> tables built for the tiles of ONE octant may be applied with n_mirrors = 8.
 *   gs_specular_tiles_apply : forward  dst[o] = sum_i g(o,i) src[i] scale[i] / sum_i g(o,i) scale[i]      (src = cubemap level [6,R,R,3],
 *                                      scale[6 R^2] = pixel_area / 4)
 *                             backward dst[i] = out_scale[i] sum_o g(o,i) src[o] scale[o]                  (src = d loss / d level,
 *                                      scale = 1 / sum of the pair weights of o, out_scale = pixel_area / 4)
 *                             for the tiles [tile_begin, tile_end) of `tiles` (and their 8 reflections when n_mirrors == 8; one rank's
 *                             share when the prefilter is sharded).  One workgroup of 16 waves per (tile, reflection); 16 / nb waves
 *                             share a block.  segments[n_tiles*6][4] = {x0, y0, rows, pitch} of the source rectangle staged in LDS
 *                             per (tile, face) (pitch 0 = no rows), lds_bytes = max rows * pitch * 16. */
int gs_specular_tiles_count(int R, float roughness, float costheta_cutoff, int backward, int margin, int bw, int nb, const float* bounds,
                            const float* dir_table, const int32_t* tiles, int n_tiles, int32_t* row_counts, int32_t* extents,
                            uint64_t* pairs, void* stream);
int gs_specular_tiles_fill(int R, float roughness, float costheta_cutoff, int backward, int margin, int bw, int nb, const float* bounds,
                           const float* dir_table, const int32_t* tiles, int n_tiles, const int64_t* row_begin,
                           const int32_t* segments, int32_t* desc, float* weights, void* stream);
int gs_specular_tiles_check(int R, const float* bounds, const float* dir_table, uint64_t* out2, void* stream);
int gs_specular_tiles_apply(int R, int backward, int n_mirrors, int margin, int bw, int nb, const float* src, const float* scale,
                            const float* out_scale, const float* bounds, const int32_t* tiles, const int32_t* segments,
                            const int64_t* row_begin, const int32_t* row_counts, const int32_t* desc, const float* weights, float* dst,
                            int tile_begin, int tile_end, size_t lds_bytes, void* stream);

/* The same for SEVERAL levels in one launch (the six specular_cubemap calls of TextureCubeMap.as_splitsum,
 * rfstudio/graphics/_mesh/_texture.py:546-556, or their six backward calls): one entry per level with the arguments of
 * gs_specular_tiles_apply; the levels must not alias (each reads its own src and writes its own dst) and must share n_mirrors.
 * Workgroups are numbered level after level in the order given.  Results are bit-identical to the per-level calls. */
typedef struct GsTileLevel {
    int R, n_mirrors, margin, bw, nb, tile_begin, tile_end, reserved;
    const float* src; const float* scale; const float* out_scale; const float* bounds;
    const int32_t* tiles; const int32_t* segments; const int64_t* row_begin; const int32_t* row_counts; const int32_t* desc;
    const float* weights; float* dst; size_t lds_bytes;
} GsTileLevel;
int gs_specular_tiles_apply_multi(int n_levels, const GsTileLevel* levels, int backward, void* stream);

/* ------------------------------------------------------------------ M1: MGAdapter (mesh -> Gaussians) */
/* rfstudio/model/geosplat.py:378-472 (MGAdapter.make, default ratios): every face -> 6 flat Gaussians (two rings
 * of three).  Output rows are part-major like the reference's Splats.cat: row = part * F + face, N = 6 F.
 *   means[N,3]; scales[N,3] log-scales (third = -10); quats[N,4] wxyz; normals[N,3] interpolated shading normals.
 * faces[F,3] int64 vertex indices; vertices[V,3]; vnormals[V,3].  Opacities are the constant logit(0.99). */
int gs_mgadapter_fwd(int F, const float* vertices, const int64_t* faces, const float* vnormals,
                     float* means, float* scales, float* quats, float* normals, void* stream);
/* v_vertices[V,3], v_vnormals[V,3] are WRITTEN (zeroed by the call, then accumulated with fp32 atomics).
 * v_normals may be NULL (no gradient through the shading normals). */
int gs_mgadapter_bwd(int F, int V, const float* vertices, const int64_t* faces, const float* vnormals,
                     const float* v_means, const float* v_scales, const float* v_quats, const float* v_normals,
                     float* v_vertices, float* v_vnormals, void* stream);

/* compute_vertex_normals_(fix=True), rfstudio/graphics/_mesh/_triangle_mesh.py:588-613: area-weighted vertex normals.
 * raw[V,3] = un-normalised sums (saved for the backward); |raw| <= 1e-10 -> (0,0,1) with zero gradient.
 * bwd: v_raw[V,3] scratch; v_vertices[V,3] written, or accumulated into when accumulate != 0 (so that it can be
 * chained onto gs_mgadapter_bwd's v_vertices). */
int gs_vertex_normals_fwd(int F, int V, const float* vertices, const int64_t* faces, float* raw, float* vnormals,
                          void* stream);
int gs_vertex_normals_bwd(int F, int V, const float* vertices, const int64_t* faces, const float* raw,
                          const float* v_vnormals, float* v_raw, float* v_vertices, int accumulate, void* stream);

/* ------------------------------------------------------------------ L1: loss side (per view) ------- */
/* rfstudio/trainer/geosplat_trainer.py:171-195 for one view, value AND gradient in one call:
 *   x = rgb + (1-alpha) train_bg;  y = lin(gt) mask + (1-mask) train_bg;  mask = gt alpha
 *   loss = ssim_lambda (1 - SSIM(y,x)) + (1 - ssim_lambda) L1(x,y) + mask_weight mean((mask-alpha)^2)
 * SSIM = torchmetrics structural_similarity_index_measure defaults (11x11 Gaussian, sigma 1.5; loss/photometric_loss.py:73-112).
 *   rgb[H,W,3] linear (the path's tone-mapped output), alpha[H,W], gt_rgba[H,W,4] (sRGB colours when gt_is_srgb != 0:
 *   srgb2rgb of graphics/_images.py:287-311 is applied inside), train_bg[H,W,3] (the caller's torch.rand_like),
 *   metric_bg[3] device or NULL (background of the sRGB PSNR metric, :191-195).
 *   out[6] device = {loss, 1-ssim, l1, mask_mse, mse_srgb, psnr_srgb}  (unscaled)
 *   v_rgb[H,W,3], v_alpha[H,W] = grad_scale * d loss / d(rgb, alpha)  (both NULL: value only).
 * W, H must exceed 10.  Deterministic (fixed-order reduction). */
size_t gs_photo_loss_ws_bytes(int W, int H);
int gs_photo_loss(int W, int H, const float* rgb, const float* alpha, const float* gt_rgba, int gt_is_srgb,
                  const float* train_bg, const float* metric_bg, float ssim_lambda, float mask_weight,
                  float grad_scale, float* out, float* v_rgb, float* v_alpha, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ F1: hash-grid encoding --------- */
/* HashEncoding, `backend='torch'` semantics (rfstudio/model/components/encoding.py:187-229): every level hashed,
 * x in [-1,1]^3 -> x/2+1/2, scalings[l] = floor(min_res * growth^l) (host array of L floats, computed by the caller
 * exactly as encoding.py:124-132 does), corners {ceil, floor}, hash = (x ^ y*2654435761 ^ z*805459861) mod 2^log2_T.
 *   x[N,3]; table[L * 2^log2_T, F]; out[N, L*F].  F must be 2 (rfstudio/model/geosplat.py:485-518).
 * bwd: v_table (+)= table_grad_scale * d/dtable (written, or added to when accumulate != 0; table_grad_scale
 *      carries the reference's grad_scaling trick, encoding.py:231-240), v_x[N,3] written (may be NULL). */
int gs_hashgrid_fwd(int N, int L, int F, int log2_T, const float* scalings_host, const float* x, const float* table,
                    float* out, void* stream);
size_t gs_hashgrid_bwd_ws_bytes(int N, int L, int F);         /* level-major copy of v_out */
int gs_hashgrid_bwd(int N, int L, int F, int log2_T, const float* scalings_host, const float* x, const float* table,
                    const float* v_out, float table_grad_scale, float* v_table, int accumulate, float* v_x,
                    void* ws /* with it: atomic-free LDS-slab kernel; NULL: per-point kernel with fp32 atomics */,
                    size_t ws_bytes, void* stream);

/* gs_hashgrid_bwd with the table gradient accumulated in 64-bit FIXED POINT over binned points: deterministic (integer
 * sums), and faster -- LDS float atomics retire one lane per ~3 cycles on gfx950, integer ones 18-30x as many.  One power
 * of two per level, from that level's max |v_out| (reduced on the device), scales the fixed point.  Arguments as
 * gs_hashgrid_bwd; ws: gs_hashgrid_bwd_fixed_ws_bytes(N, L, F, log2_T) bytes (0: table size outside 2^15..2^19 rows). */
size_t gs_hashgrid_bwd_fixed_ws_bytes(int N, int L, int F, int log2_T);
int gs_hashgrid_bwd_fixed(int N, int L, int F, int log2_T, const float* scalings_host, const float* x, const float* table,
                          const float* v_out, float table_grad_scale, float* v_table, int accumulate, float* v_x, void* ws,
                          size_t ws_bytes, void* stream);

/* Weight gradient of one bias-free linear layer of the field's MLPs (rfstudio/nn/mlp.py:126-145; widths <= 32):
 *   dW[O,I] (+)= scale * dY[N,O]^T X[N,I]   (row-major, contiguous; O, I in [1,32]).  fp32 matrix unit, exact f32;
 *   deterministic (partials summed in a fixed order).  ws: gs_mlp_wgrad_ws_bytes(N). */
size_t gs_mlp_wgrad_ws_bytes(int64_t N);
int gs_mlp_wgrad(int64_t N, int O, int I, const float* dY, const float* X, float scale, float* dW, int accumulate,
                 void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ X1: FlexiCubes extraction ------ */
/* FlexiCubes.dual_marching_cubes(grad_func=None) + compute_entropy on the regular grid of FlexiCubes.from_resolution
 * (rfstudio/graphics/_mesh/_flexicubes.py:397-457, 559-713, 715-802; caller rfstudio/model/geosplat.py:751-769).
 * Grid: cube n = i0 + R0 (i1 + R1 i2); grid vertex id = i0 + (R0+1) (i1 + (R1+1) i2); cube corner k adds bit j of k to ij
 * (exactly the `indices` from_resolution builds, which are therefore not passed).
 *   vertices[Vg,3], sdf[Vg]; raw alpha[C,8] / beta[C,12] / gamma[C] or NULL (the tanh / sigmoid * weight_scale
 *   activations of :608-624 are applied inside); sdf_eps < 0 means None.
 * Step 1  gs_flexicubes_count: classifies cubes and grid edges into `ws`, writes counts[8] (device int64) =
 *         {surface cubes N, dual vertices Q, (patch,edge) pairs K, surface edges E, quads, flipped quads, 0, 0}.
 *         The caller reads them (the reference's own .item() syncs) and allocates the outputs.
 * Step 2  gs_flexicubes_fwd: out_vertices[Q + quads, 3] (dual vertices, then one centre vertex per quad),
 *         out_faces[4 * quads, 3] int64, L_dev[K] -- vertex / face / row numbering identical to the reference's.
 * Step 3  gs_flexicubes_bwd: cotangents v_out_vertices[Q+quads,3], v_L_dev[K] (or NULL) -> g_vertices[Vg,3], g_sdf[Vg],
 *         g_alpha[C,8], g_beta[C,12], g_gamma[C] (w.r.t. the RAW weights; NULL iff the weight was NULL), all overwritten.
 *         g_vd_scratch: 3*Q floats.  `ws` must still hold what step 1 left for the same sdf.
 * gs_flexicubes_entropy_fwd/bwd: compute_entropy (:715-725) of the same sdf (needs step 1's `ws`); out[1]; bwd adds
 *         v_out[0] * d/dsdf into g_sdf (overwrites when accumulate == 0). */
size_t gs_flexicubes_ws_bytes(int R0, int R1, int R2);
int gs_flexicubes_count(int R0, int R1, int R2, const float* sdf, void* ws, size_t ws_bytes, int64_t* counts, void* stream);
int gs_flexicubes_fwd(int R0, int R1, int R2, const float* vertices, const float* sdf, const float* alpha,
                      const float* beta, const float* gamma, float weight_scale, float sdf_eps, const void* ws,
                      size_t ws_bytes, int64_t Q, int64_t num_quads, int64_t K, float* out_vertices, int64_t* out_faces,
                      float* L_dev, void* stream);
int gs_flexicubes_bwd(int R0, int R1, int R2, const float* vertices, const float* sdf, const float* alpha,
                      const float* beta, const float* gamma, float weight_scale, float sdf_eps, const void* ws,
                      size_t ws_bytes, int64_t Q, int64_t num_quads, int64_t K, const float* out_vertices,
                      const float* v_out_vertices, const float* v_L_dev, float* g_vd_scratch, float* g_vertices,
                      float* g_sdf, float* g_alpha, float* g_beta, float* g_gamma, void* stream);
int gs_flexicubes_entropy_fwd(int R0, int R1, int R2, const float* sdf, void* ws, size_t ws_bytes, float* out, void* stream);
int gs_flexicubes_entropy_bwd(int R0, int R1, int R2, const float* sdf, const void* ws, size_t ws_bytes, const float* v_out,
                              float* g_sdf, int accumulate, void* stream);

/* ------------------------------------------------------------------ fused front / tail of a view (the engine's path) ---------- */
/* Shading fused AHEAD of the projection (north star; reference semantics rfstudio/model/geosplat.py:80-128 followed by the
 * rasterization call at rfstudio/model/gsplat.py:334-355): one launch projects every Gaussian (gs_project_fwd's arithmetic),
 * shades the visible ones (gs_shade_fwd's arithmetic) and writes, in packed order (ascending Gaussian index),
 *   vis_records [N,16] f32 : {mx, my, a/2, b | c/2, opacity*comp, hx, hy | r, g, b, 0 | comp, bits(Gaussian index), depth, bits(radius)}
 *                            -- gs_raster_prepare_vis(_cap) consumes it as is, gs_tail_bwd reads the fourth quarter;
 *   depth_keys [N] u32     : the binning's depth key.  key_bits == 32: the depth's float bits.  key_bits == 24: bits - key_base,
 *                            valid while key_base <= bits < key_base + 2^24 for every visible Gaussian -- otherwise
 *                            status4[3] is set to 1 (int64[4], caller-zeroed, sticky; words 0..2 as for gs_isect_bin_cap) and the
 *                            key is clamped: memory-safe, wrong order, reported;
 *   tile_rects [N,2] u32   : (x0 | y0 << 16, x1 | y1 << 16), the tile rectangle of gs_project_fwd's tiles_per_gauss;
 *   tile_counts [tiles] u32: intersections per tile (zeroed by the call; nullable, and not written for more than 8 192 tiles):
 *                            gs_isect_bin_front turns them into the tile offsets;
 *   counts4 [4] i64        : {V, I, 0xffffffff - min depth bits, max depth bits} (zeroed by the call).
 * No other per-visible array exists on this path.  Scratch: gs_front_ws_bytes(N), zeroed by the call.
 * tight_tiles != 0: the tile rectangle is gsplat's square intersected with the axis-aligned extent of {alpha >= 1/255} (the extents
 * hx, hy of the record, with which the compositor culls its quadrants anyway): every pixel composites exactly the same Gaussians in
 * the same order, I is smaller (a caller that needs gsplat's `meta` passes 0).
 * Two partial forms (same packed order, same counts): vis_records == NULL = GEOMETRY ONLY (keys, rectangles, tile counts, slots;
 * nothing is shaded, env may be NULL) -- the engine bins its first view with it while the prefilter is still computing the
 * pyramid -- and depth_keys == tile_rects == NULL = records only. */
size_t gs_front_ws_bytes(int N);
int gs_front_fwd(int N, const float* means, const float* quats, const float* scales, const float* opacities,
                 const float* normals, const float* kd, const float* ks, const float* viewmat, const float* K,
                 const float* cam_pos, float min_roughness, float max_metallic, int mode, const GsEnv* env,
                 int W, int H, int tile_size, float eps2d, float near_plane, float far_plane, float radius_clip,
                 uint32_t key_base, int key_bits, float* vis_records, uint32_t* depth_keys, uint32_t* tile_rects,
                 uint32_t* tile_counts, int32_t* packed_index /* [N], nullable: packed slot of every Gaussian or -1 */,
                 int tight_tiles, int64_t* counts4, int64_t* status4, void* ws, size_t ws_bytes, void* stream);
/* Binning from gs_front_fwd's outputs: flatten_ids_sorted in (tile, depth, packed index) order and the tile offsets, bit-identical
 * to gs_isect_bin_tiles_cap + gs_isect_offsets_tiles_cap.  counts_dev == NULL: V_cap / n_isects_cap are the exact counts; otherwise
 * the capacity protocol of gs_isect_bin_cap (status_dev int64[4]).  key_bits as passed to gs_front_fwd. */
size_t gs_isect_bin_front_ws_bytes(int V, int64_t n_isects, int tile_w, int tile_h);
int gs_isect_bin_front(int V_cap, const uint32_t* depth_keys, const uint32_t* tile_rects, const uint32_t* tile_counts,
                       const int64_t* counts_dev, int64_t n_isects_cap, int key_bits, int tile_w, int tile_h, int32_t* flatten_ids_sorted,
                       int32_t* isect_offsets, void* ws, size_t ws_bytes, int64_t* status_dev, void* stream);
/* Projection backward fused with the shading backward (gs_project_bwd + gs_shade_bwd in one launch, one thread per VISIBLE
 * Gaussian): reads vis_records and the compositor's gradient records v_packed [V, rec_stride] ({xy, conic(3), opacity, rgb}),
 * ADDS into v_means [N,3], v_quats [N,4], v_scales [N,3] (gradient w.r.t. the scales passed in), v_opacities [N], v_normals,
 * v_kd [N,3], v_ks [N,2] and into the texel gradients env_grad.  counts_dev == NULL: V_cap is the count. */
int gs_tail_bwd(int V_cap, const int64_t* counts_dev, const float* means, const float* quats, const float* scales,
                const float* opacities, const float* normals, const float* kd, const float* ks, const float* viewmat,
                const float* K, const float* cam_pos, float min_roughness, float max_metallic, int mode, const GsEnv* env,
                int W, int H, float eps2d, const float* vis_records, const float* v_packed, int rec_stride,
                float* v_means, float* v_quats, float* v_scales, float* v_opacities, float* v_normals, float* v_kd, float* v_ks,
                const GsEnvGrad* env_grad, void* priv_ws, size_t priv_ws_bytes, void* stream);
/* The tails of ALL views of a step in one call (two launches: shading half, projection half): up to eight adjacent lanes own one
 * Gaussian, one view each, the 19 parameter gradients are summed over the lanes and stored once (accumulate == 0: plain stores
 * into v_*, every Gaussian written; != 0: added).  Per view: what gs_front_fwd wrote (vis_records, packed_index) and the
 * compositor's v_packed [V, rec_stride].  The gradients of gs_tail_bwd summed over the views, to summation order (the views as a
 * tree, the colour cotangent contracted into the cube fetch: tests/test_gpu_front.py, 1e-5).  More than eight views: further
 * launches add to the first. */
typedef struct GsTailView {
    const float* viewmat; const float* K; const float* cam_pos;            /* device: [4,4], [3,3], [3] */
    const float* vis_records; const float* v_packed; const int32_t* packed_index;
    int W, H;
} GsTailView;
int gs_tail_bwd_multi(int N, int n_views, const GsTailView* views /*host array*/, const float* means, const float* quats,
                      const float* scales, const float* opacities, const float* normals, const float* kd, const float* ks,
                      float min_roughness, float max_metallic, int mode, const GsEnv* env, float eps2d, int rec_stride,
                      float* v_means, float* v_quats, float* v_scales, float* v_opacities, float* v_normals, float* v_kd, float* v_ks,
                      int accumulate, const GsEnvGrad* env_grad, void* priv_ws, size_t priv_ws_bytes, void* stream);
/* The two halves of gs_tail_bwd_multi as separate calls (bit 0 of `parts`: the S1-S3 shading backward -- writes v_means (its
 * view-direction part), v_normals, v_kd, v_ks and the texel gradients; bit 1: the A7 projection backward -- ADDS its part to
 * v_means, writes v_quats, v_scales, v_opacities), so that the caller can put them on different streams: the projection half has
 * to follow the shading half of the same call (v_means), nothing else orders them.  parts == 3 is gs_tail_bwd_multi.
 * Bit 2 (with bit 0): a BACKGROUND launch -- the shading half on half of the CUs (128 persistent workgroups), for a tail that is
 * not the last of its step and runs beside the compositor of the following views; same results, same order of the sums per
 * Gaussian (texel gradients are float atomics in every form). */
int gs_tail_bwd_multi_parts(int parts, int N, int n_views, const GsTailView* views /*host array*/, const float* means, const float* quats,
                            const float* scales, const float* opacities, const float* normals, const float* kd, const float* ks,
                            float min_roughness, float max_metallic, int mode, const GsEnv* env, float eps2d, int rec_stride,
                            float* v_means, float* v_quats, float* v_scales, float* v_opacities, float* v_normals, float* v_kd, float* v_ks,
                            int accumulate, const GsEnvGrad* env_grad, void* priv_ws, size_t priv_ws_bytes, void* stream);
/* Optional scratch of gs_tail_bwd / gs_tail_bwd_multi: eight XCD-private copies of the mid-sized specular levels (64^2, 128^2 of a 512^2 pyramid), whose
 * texel atomics then resolve in the XCD's own L2 instead of at the memory side of the fabric.  The caller zeroes priv_ws
 * (gs_tail_priv_ws_bytes) once, every gs_tail_bwd of a step ADDS into it, gs_tail_priv_reduce folds it into env_grad once.
 * priv_ws == NULL: every texel gradient goes straight into env_grad. */
size_t gs_tail_priv_ws_bytes(const GsEnv* env /*host*/, int mode);
int gs_tail_priv_reduce(const GsEnv* env /*host*/, int mode, const void* priv_ws, size_t priv_ws_bytes, const GsEnvGrad* env_grad,
                        void* stream);

/* Chain rule of the per-step activations of GSplatter.render_rgba (rfstudio/model/gsplat.py:336-339: `scales.exp()`,
 * `sigmoid(opacities)`) applied once to gradients accumulated over the views of a step:
 *   v_scales[N,3] = g_scales_act * scales_act,   v_opacities[N] = (g_opac_act * opac_act) * (1 - opac_act). */
int gs_activation_chain(int64_t N, const float* g_scales_act, const float* scales_act, const float* g_opac_act, const float* opac_act,
                        float* v_scales, float* v_opacities, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOSPLAT_HIP_H */
