/*
 * gs_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the rasterizer half of the GeoSplatting hot path:
 * the arithmetic that `gsplat.rasterization(...)` performs when it is called by
 * the reference at rfstudio/model/gsplat.py:334-355 (packed=True,
 * rasterize_mode='antialiased', tile_size=16, near=0.01, far=1e10).
 *
 * PARITY UNPINNED: the arithmetic lives in the third-party package gsplat
 * (pinned `gsplat~=1.4.0`, /root/reference/pyproject.toml:24) whose source is
 * absent from /root/reference; the reference holds no golden vectors for it
 * (SURVEY.md section 4 / 8c).  This file restates the published gsplat-1.4
 * algorithm (stage names A1..A7 follow SURVEY.md section 8a) and is anchored on
 * the reference's call sites only.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (geosplatting_amd/) never does.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp).
 * All floating point is IEEE fp32, one rounding per written operation (no
 * contraction) except where fmaf() is spelled out; per-Gaussian gradient
 * accumulators in the backward compositor are double (documented below).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GSO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* A1: fully fused projection, packed mode (gsplat 1.4 fully_fused_projection
 * with packed=True, calc_compensations=True; called through
 * rfstudio/model/gsplat.py:334).  One camera (reference asserts
 * cameras.shape == (1,), rfstudio/model/gsplat.py:293).                      */

typedef struct {
    float R[9];   /* world->camera rotation, row-major */
    float t[3];
    float fx, fy, cx, cy;
    int W, H;
} gso_cam;

static void cam_from(const float* viewmat, const float* K, int W, int H, gso_cam* c)
{
    /* viewmat: row-major 4x4 as produced by Cameras.view_matrix
     * (rfstudio/graphics/_cameras.py:299-314); K: row-major 3x3 from
     * Cameras.intrinsic_matrix (rfstudio/graphics/_cameras.py:289-297). */
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c->R[i * 3 + j] = viewmat[i * 4 + j];
        c->t[i] = viewmat[i * 4 + 3];
    }
    c->fx = K[0]; c->fy = K[4]; c->cx = K[2]; c->cy = K[5];
    c->W = W; c->H = H;
}

static void quat_to_rotmat(const float* q, float* R /*row-major*/)
{
    float w = q[0], x = q[1], y = q[2], z = q[3];      /* wxyz (rfstudio/graphics/math.py:246-278) */
    float n2 = ((x * x + y * y) + z * z) + w * w;
    float inv = 1.0f / sqrtf(n2);
    x *= inv; y *= inv; z *= inv; w *= inv;
    float x2 = x * x, y2 = y * y, z2 = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.0f - 2.0f * (y2 + z2); R[1] = 2.0f * (xy - wz);        R[2] = 2.0f * (xz + wy);
    R[3] = 2.0f * (xy + wz);        R[4] = 1.0f - 2.0f * (x2 + z2); R[5] = 2.0f * (yz - wx);
    R[6] = 2.0f * (xz - wy);        R[7] = 2.0f * (yz + wx);        R[8] = 1.0f - 2.0f * (x2 + y2);
}

/* C = A * B, 3x3 row-major, k summed in order 0,1,2 */
static void mat3_mul(const float* A, const float* B, float* C)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j]) + A[i * 3 + 2] * B[2 * 3 + j];
}
/* C = A * B^T */
static void mat3_mul_bt(const float* A, const float* B, float* C)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i * 3 + 0] * B[j * 3 + 0] + A[i * 3 + 1] * B[j * 3 + 1]) + A[i * 3 + 2] * B[j * 3 + 2];
}

typedef struct {
    int valid;
    int radius;
    float mean2d[2];
    float depth;
    float conic[3];
    float comp;
    /* intermediates reused by the backward */
    float mean_c[3];
    float covar_c[9];
    float covar[9];
    float Rq[9];
} gso_proj;

static void project_one(const gso_cam* c, const float* mean, const float* quat, const float* scale,
                        float eps2d, float near_plane, float far_plane, float radius_clip, gso_proj* o)
{
    o->valid = 0; o->radius = 0;
    /* camera-space mean */
    float mc[3];
    for (int i = 0; i < 3; ++i)
        mc[i] = ((c->R[i * 3 + 0] * mean[0] + c->R[i * 3 + 1] * mean[1]) + c->R[i * 3 + 2] * mean[2]) + c->t[i];
    o->mean_c[0] = mc[0]; o->mean_c[1] = mc[1]; o->mean_c[2] = mc[2];
    if (mc[2] < near_plane || mc[2] > far_plane) return;

    /* covariance from quaternion and (already exp'ed) scale */
    float M[9];
    quat_to_rotmat(quat, o->Rq);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i * 3 + j] = o->Rq[i * 3 + j] * scale[j];
    mat3_mul_bt(M, M, o->covar);
    float T1[9];
    mat3_mul(c->R, o->covar, T1);
    mat3_mul_bt(T1, c->R, o->covar_c);
    const float* Cc = o->covar_c;

    /* perspective projection with fov clamp */
    float x = mc[0], y = mc[1], z = mc[2];
    float W = (float)c->W, H = (float)c->H;
    float tan_fovx = 0.5f * W / c->fx;
    float tan_fovy = 0.5f * H / c->fy;
    float lim_x_pos = (W - c->cx) / c->fx + 0.3f * tan_fovx;
    float lim_x_neg = c->cx / c->fx + 0.3f * tan_fovx;
    float lim_y_pos = (H - c->cy) / c->fy + 0.3f * tan_fovy;
    float lim_y_neg = c->cy / c->fy + 0.3f * tan_fovy;
    float rz = 1.0f / z;
    float rz2 = rz * rz;
    float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, x * rz));
    float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, y * rz));
    float J00 = c->fx * rz, J02 = -c->fx * tx * rz2;
    float J11 = c->fy * rz, J12 = -c->fy * ty * rz2;
    /* A = J * Cc (2x3), cov2d = A * J^T */
    float A00 = J00 * Cc[0] + J02 * Cc[6], A01 = J00 * Cc[1] + J02 * Cc[7], A02 = J00 * Cc[2] + J02 * Cc[8];
    float A10 = J11 * Cc[3] + J12 * Cc[6], A11 = J11 * Cc[4] + J12 * Cc[7], A12 = J11 * Cc[5] + J12 * Cc[8];
    float c00 = A00 * J00 + A02 * J02;
    float c01 = A01 * J11 + A02 * J12;
    float c10 = A10 * J00 + A12 * J02;
    float c11 = A11 * J11 + A12 * J12;
    float m2x = c->fx * x * rz + c->cx;
    float m2y = c->fy * y * rz + c->cy;

    /* anti-aliasing blur + compensation */
    float det_orig = c00 * c11 - c01 * c10;
    c00 += eps2d; c11 += eps2d;
    float det = c00 * c11 - c01 * c10;
    float comp = sqrtf(fmaxf(0.0f, det_orig / det));
    if (det <= 0.0f) return;
    float inv_det = 1.0f / det;
    float ca = c11 * inv_det, cb = -c01 * inv_det, cc = c00 * inv_det;

    /* 3-sigma radius */
    float b = 0.5f * (c00 + c11);
    float v1 = b + sqrtf(fmaxf(0.01f, b * b - det));
    float radius = ceilf(3.0f * sqrtf(v1));
    if (radius <= radius_clip) return;
    if (m2x + radius <= 0.0f || m2x - radius >= W || m2y + radius <= 0.0f || m2y - radius >= H) return;

    o->valid = 1;
    o->radius = (int)radius;
    o->mean2d[0] = m2x; o->mean2d[1] = m2y;
    o->depth = z;
    o->conic[0] = ca; o->conic[1] = cb; o->conic[2] = cc;
    o->comp = comp;
}

/* Returns V (number of visible Gaussians).  Outputs must have capacity N.
 * Packed outputs are in ascending Gaussian index. */
GSO_API int gso_project_fwd(int N, const float* means, const float* quats, const float* scales,
                            const float* viewmat, const float* K, int W, int H,
                            float eps2d, float near_plane, float far_plane, float radius_clip,
                            int32_t* gaussian_ids, int32_t* radii, float* means2d, float* depths,
                            float* conics, float* compensations)
{
    gso_cam cam; cam_from(viewmat, K, W, H, &cam);
    uint8_t* vis = (uint8_t*)malloc((size_t)N + 1);
    gso_proj* tmp = (gso_proj*)malloc(sizeof(gso_proj) * (size_t)(N > 0 ? N : 1));
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        project_one(&cam, means + 3 * (size_t)n, quats + 4 * (size_t)n, scales + 3 * (size_t)n,
                    eps2d, near_plane, far_plane, radius_clip, &tmp[n]);
        vis[n] = (uint8_t)tmp[n].valid;
    }
    int V = 0;
    for (int n = 0; n < N; ++n) {
        if (!vis[n]) continue;
        gaussian_ids[V] = n;
        radii[V] = tmp[n].radius;
        means2d[2 * V] = tmp[n].mean2d[0]; means2d[2 * V + 1] = tmp[n].mean2d[1];
        depths[V] = tmp[n].depth;
        conics[3 * V] = tmp[n].conic[0]; conics[3 * V + 1] = tmp[n].conic[1]; conics[3 * V + 2] = tmp[n].conic[2];
        compensations[V] = tmp[n].comp;
        ++V;
    }
    free(tmp); free(vis);
    return V;
}

/* ------------------------------------------------------------------------- */
/* A2: isect_tiles.  tile_min inclusive / tile_max exclusive, clamped to the
 * tile grid; keys = (tile_id << 32) | float_bits(depth) (one camera -> camera
 * bits are zero); values = packed index.  Emission order: ascending packed
 * index, then tile row-major (y outer, x inner).                             */

static void tile_range(float mx, float my, int radius, int tile_size, int tw, int th,
                       int* x0, int* y0, int* x1, int* y1)
{
    float ts = (float)tile_size;
    float tr = (float)radius / ts;
    float tx = mx / ts, ty = my / ts;
    float fx0 = floorf(tx - tr), fy0 = floorf(ty - tr);
    float fx1 = ceilf(tx + tr), fy1 = ceilf(ty + tr);
    /* (uint32_t) cast of a negative float saturates to 0 on the GPU; then min with the grid size */
    *x0 = fx0 < 0.0f ? 0 : (fx0 > (float)tw ? tw : (int)fx0);
    *y0 = fy0 < 0.0f ? 0 : (fy0 > (float)th ? th : (int)fy0);
    *x1 = fx1 < 0.0f ? 0 : (fx1 > (float)tw ? tw : (int)fx1);
    *y1 = fy1 < 0.0f ? 0 : (fy1 > (float)th ? th : (int)fy1);
}

GSO_API int64_t gso_isect_count(int V, const float* means2d, const int32_t* radii, int tile_size,
                                int tile_w, int tile_h, int32_t* tiles_per_gauss)
{
    int64_t total = 0;
    for (int v = 0; v < V; ++v) {
        int x0, y0, x1, y1;
        if (radii[v] <= 0) { tiles_per_gauss[v] = 0; continue; }
        tile_range(means2d[2 * v], means2d[2 * v + 1], radii[v], tile_size, tile_w, tile_h, &x0, &y0, &x1, &y1);
        int n = (x1 - x0) * (y1 - y0);
        tiles_per_gauss[v] = n;
        total += n;
    }
    return total;
}

GSO_API void gso_isect_emit(int V, const float* means2d, const int32_t* radii, const float* depths,
                            int tile_size, int tile_w, int tile_h, int64_t* isect_ids, int32_t* flatten_ids)
{
    int64_t cur = 0;
    for (int v = 0; v < V; ++v) {
        int x0, y0, x1, y1;
        if (radii[v] <= 0) continue;
        tile_range(means2d[2 * v], means2d[2 * v + 1], radii[v], tile_size, tile_w, tile_h, &x0, &y0, &x1, &y1);
        int32_t dbits; memcpy(&dbits, &depths[v], 4);
        int64_t depth_enc = (int64_t)(uint32_t)dbits;
        for (int i = y0; i < y1; ++i)
            for (int j = x0; j < x1; ++j) {
                int64_t tile_id = (int64_t)i * tile_w + j;
                isect_ids[cur] = (tile_id << 32) | depth_enc;
                flatten_ids[cur] = v;
                ++cur;
            }
    }
}

/* ------------------------------------------------------------------------- */
/* A3: stable ascending sort of (isect_ids, flatten_ids) -- semantics of
 * cub::DeviceRadixSort::SortPairs.  LSD radix, 8 bits per pass, 8 passes.    */
GSO_API void gso_sort_pairs(int64_t n, int64_t* keys, int32_t* vals)
{
    if (n <= 1) return;
    uint64_t* k0 = (uint64_t*)keys;
    uint64_t* k1 = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
    int32_t* v0 = vals;
    int32_t* v1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    for (int pass = 0; pass < 8; ++pass) {
        size_t cnt[257]; memset(cnt, 0, sizeof(cnt));
        int sh = pass * 8;
        for (int64_t i = 0; i < n; ++i) cnt[((k0[i] >> sh) & 0xff) + 1]++;
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; ++i) {
            size_t p = cnt[(k0[i] >> sh) & 0xff]++;
            k1[p] = k0[i]; v1[p] = v0[i];
        }
        uint64_t* tk = k0; k0 = k1; k1 = tk;
        int32_t* tv = v0; v0 = v1; v1 = tv;
    }
    /* 8 passes -> result is back in the caller's buffers */
    free(k1); free(v1);
}

/* A4: isect_offset_encode: offsets[t] = first sorted position whose tile >= t */
GSO_API void gso_isect_offsets(int64_t n_isects, const int64_t* isect_ids, int n_tiles, int32_t* offsets)
{
    int64_t p = 0;
    for (int t = 0; t < n_tiles; ++t) {
        while (p < n_isects && (int64_t)(((uint64_t)isect_ids[p]) >> 32) < t) ++p;
        offsets[t] = (int32_t)p;
    }
}

/* ------------------------------------------------------------------------- */
/* A5: rasterize_to_pixels forward.  Pixel centre (j+0.5, i+0.5); per pixel
 * walk the tile's sorted range front to back.
 *   sigma = 0.5*(a dx^2 + c dy^2) + b dx dy  (spelled with fmaf, see below)
 *   alpha = min(0.999, opac * exp(-sigma)); skip if sigma<0 or alpha<1/255
 *   stop (without compositing) when T*(1-alpha) <= 1e-4
 * colors/opacities are PACKED ([V,D], [V]; opacity already * compensation).  */

static inline float gso_sigma(float a, float b, float c, float dx, float dy)
{
    /* canonical evaluation order shared with the HIP kernel (explicit FMAs) */
    float t0 = (0.5f * a) * dx;
    float t1 = (0.5f * c) * dy;
    float t2 = b * dx;
    return fmaf(t0, dx, fmaf(t1, dy, t2 * dy));
}

/* exp(-sigma) in ONE canonical operation order shared with the HIP compositor (gs_common.h gs_exp_neg): the upstream
 * kernel calls the hardware's approximate exponential (__expf), whose bits no CPU reproduces; spelling the function
 * out -- 2^(y) with y = -sigma*log2(e), n = rint(y), a degree-6 polynomial for 2^(y-n) in explicit FMAs, ldexp -- makes
 * alpha, the transmittance chain, every skip / stop decision and therefore the image and last_ids BIT-IDENTICAL
 * between this oracle and the GPU (max relative error of the function 8e-8, the size of __expf's own error). */
static inline float gso_exp_neg(float sigma)
{
    float y = sigma * -1.44269504f;
    float yc = fminf(fmaxf(y, -126.0f), 126.0f);   /* keeps rint / ldexp in range; NaN -> -126 */
    float n = rintf(yc);                        /* round to nearest even */
    float f = yc - n;                           /* exact */
    float p = 0x1.41a6fep-13f;
    p = fmaf(p, f, 0x1.5f44f0p-10f);
    p = fmaf(p, f, 0x1.3b2dfep-7f);
    p = fmaf(p, f, 0x1.c6aed6p-5f);
    p = fmaf(p, f, 0x1.ebfbdap-3f);
    p = fmaf(p, f, 0x1.62e430p-1f);
    p = fmaf(p, f, 1.0f);
    float r = ldexpf(p, (int)n);
    return (y >= -125.0f) ? r : 0.0f;           /* underflow (and NaN): such a pair is far below 1/255 anyway */
}

/* Test hook for the canonical exponential: over every float whose bit pattern lies in [lo_bits, hi_bits] (sigma >= 0)
 *   max_rel   = max |gso_exp_neg(sigma) - exp(-sigma)| / exp(-sigma) against the float64 libm exponential,
 *   checksum  = sum_i bits(gso_exp_neg(sigma_i)) * (2 i + 1)  mod 2^64   (order-independent; the HIP self-test
 *               gs_selftest_exp computes the same sum from gs_exp_neg, so equal sums <=> equal bits on the whole range). */
GSO_API void gso_exp_neg_check(uint32_t lo_bits, uint32_t hi_bits, double* max_rel, uint64_t* checksum)
{
    const int64_t n = (int64_t)hi_bits - (int64_t)lo_bits + 1;
    double worst = 0.0;
    uint64_t sum = 0;
#pragma omp parallel
    {
        double w = 0.0;
        uint64_t s = 0;
#pragma omp for schedule(static) nowait
        for (int64_t i = 0; i < n; ++i) {
            union { uint32_t u; float f; } in, out;
            in.u = lo_bits + (uint32_t)i;
            out.f = gso_exp_neg(in.f);
            s += (uint64_t)out.u * (2ull * (uint64_t)i + 1ull);
            const double ref = exp(-(double)in.f);
            const double e = fabs((double)out.f - ref) / ref;
            if (e > w) w = e;
        }
#pragma omp critical
        { if (w > worst) worst = w; sum += s; }
    }
    if (max_rel) *max_rel = worst;
    if (checksum) *checksum = sum;
}

GSO_API void gso_raster_fwd(int W, int H, int tile_size, int D,
                            const float* means2d, const float* conics, const float* opacities,
                            const float* colors, const float* background /* nullable [D] */,
                            int64_t n_isects, const int32_t* offsets, const int32_t* flatten_ids,
                            float* render /*[H,W,D]*/, float* alphas /*[H,W]*/, int32_t* last_ids /*[H,W]*/,
                            uint8_t* ambiguous /* nullable [H,W]: 1 if a threshold test was within 1e-5 rel */,
                            int64_t* pair_count /* nullable [2]: evaluated (pixel,gaussian) pairs, composited pairs */)
{
    int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
    int n_tiles = tw * th;
    int64_t pairs = 0, composited = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : pairs, composited)
    for (int tile = 0; tile < n_tiles; ++tile) {
        int ty = tile / tw, tx = tile % tw;
        int64_t start = offsets[tile];
        int64_t end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];
        float pix_out[64];
        for (int i = ty * tile_size; i < (ty + 1) * tile_size && i < H; ++i)
            for (int j = tx * tile_size; j < (tx + 1) * tile_size && j < W; ++j) {
                float px = (float)j + 0.5f, py = (float)i + 0.5f;
                float T = 1.0f;
                int32_t cur_idx = 0;
                int amb = 0;
                for (int k = 0; k < D; ++k) pix_out[k] = 0.0f;
                for (int64_t idx = start; idx < end; ++idx) {
                    int g = flatten_ids[idx];
                    float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
                    float sigma = gso_sigma(conics[3 * g], conics[3 * g + 1], conics[3 * g + 2], dx, dy);
                    float alpha = fminf(0.999f, opacities[g] * gso_exp_neg(sigma));
                    ++pairs;
                    if (fabsf(alpha - 1.0f / 255.0f) < 1e-5f * (1.0f / 255.0f)) amb = 1;
                    if (sigma < 0.0f || alpha < 1.0f / 255.0f) continue;
                    float next_T = T * (1.0f - alpha);
                    if (fabsf(next_T - 1e-4f) < 1e-5f * 1e-4f) amb = 1;
                    if (next_T <= 1e-4f) break;
                    float vis = alpha * T;
                    ++composited;
                    for (int k = 0; k < D; ++k) pix_out[k] = fmaf(colors[(size_t)g * D + k], vis, pix_out[k]);
                    cur_idx = (int32_t)idx;
                    T = next_T;
                }
                size_t pid = (size_t)i * W + j;
                alphas[pid] = 1.0f - T;
                for (int k = 0; k < D; ++k)
                    render[pid * D + k] = background ? fmaf(T, background[k], pix_out[k]) : pix_out[k];
                last_ids[pid] = cur_idx;
                if (ambiguous) ambiguous[pid] = (uint8_t)amb;
            }
    }
    if (pair_count) { pair_count[0] = pairs; pair_count[1] = composited; }
}

/* ------------------------------------------------------------------------- */
/* A6: rasterize_to_pixels backward (stored state: alphas, last_ids).  Walks
 * each pixel back to front from last_ids.  Per-Gaussian accumulators are
 * double so that the oracle is summation-order independent (the GPU path and
 * upstream both accumulate with fp32 atomics in arbitrary order).            */
GSO_API void gso_raster_bwd(int W, int H, int tile_size, int D,
                            const float* means2d, const float* conics, const float* opacities,
                            const float* colors, const float* background,
                            int64_t n_isects, const int32_t* offsets, const int32_t* flatten_ids,
                            const float* alphas, const int32_t* last_ids,
                            const float* v_render, const float* v_alphas, int V,
                            float* v_means2d, float* v_conics, float* v_colors, float* v_opacities)
{
    int tw = (W + tile_size - 1) / tile_size, th = (H + tile_size - 1) / tile_size;
    int n_tiles = tw * th;
    double* acc = (double*)calloc((size_t)V * (size_t)(6 + D) + 1, sizeof(double));
    /* layout per Gaussian: [xy(2), conic(3), opacity(1), color(D)] */
    const int S = 6 + D;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < n_tiles; ++tile) {
        int ty = tile / tw, tx = tile % tw;
        int64_t start = offsets[tile];
        int64_t end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];
        if (end <= start) continue;
        float buffer[64], vrc[64];
        for (int i = ty * tile_size; i < (ty + 1) * tile_size && i < H; ++i)
            for (int j = tx * tile_size; j < (tx + 1) * tile_size && j < W; ++j) {
                size_t pid = (size_t)i * W + j;
                float px = (float)j + 0.5f, py = (float)i + 0.5f;
                float T_final = 1.0f - alphas[pid];
                float T = T_final;
                float vra = v_alphas[pid];
                for (int k = 0; k < D; ++k) { buffer[k] = 0.0f; vrc[k] = v_render[pid * D + k]; }
                int64_t bin_final = last_ids[pid];
                if (bin_final >= end) bin_final = end - 1;
                for (int64_t idx = bin_final; idx >= start; --idx) {
                    int g = flatten_ids[idx];
                    float ca = conics[3 * g], cb = conics[3 * g + 1], cc = conics[3 * g + 2];
                    float opac = opacities[g];
                    float dx = means2d[2 * g] - px, dy = means2d[2 * g + 1] - py;
                    float sigma = gso_sigma(ca, cb, cc, dx, dy);
                    float vis = gso_exp_neg(sigma);
                    float alpha = fminf(0.999f, opac * vis);
                    if (sigma < 0.0f || alpha < 1.0f / 255.0f) continue;
                    float ra = 1.0f / (1.0f - alpha);
                    T *= ra;
                    float fac = alpha * T;
                    float v_alpha = 0.0f;
                    const float* col = colors + (size_t)g * D;
                    double* a = acc + (size_t)g * S;
                    for (int k = 0; k < D; ++k) {
                        v_alpha += (col[k] * T - buffer[k] * ra) * vrc[k];
                        double vc = (double)(fac * vrc[k]);
#pragma omp atomic
                        a[6 + k] += vc;
                    }
                    v_alpha += T_final * ra * vra;
                    if (background) {
                        float accum = 0.0f;
                        for (int k = 0; k < D; ++k) accum += background[k] * vrc[k];
                        v_alpha += -T_final * ra * accum;
                    }
                    if (opac * vis <= 0.999f) {
                        float v_sigma = -opac * vis * v_alpha;
                        double g_ca = (double)(0.5f * v_sigma * dx * dx);
                        double g_cb = (double)(v_sigma * dx * dy);
                        double g_cc = (double)(0.5f * v_sigma * dy * dy);
                        double g_x = (double)(v_sigma * (ca * dx + cb * dy));
                        double g_y = (double)(v_sigma * (cb * dx + cc * dy));
                        double g_o = (double)(vis * v_alpha);
#pragma omp atomic
                        a[0] += g_x;
#pragma omp atomic
                        a[1] += g_y;
#pragma omp atomic
                        a[2] += g_ca;
#pragma omp atomic
                        a[3] += g_cb;
#pragma omp atomic
                        a[4] += g_cc;
#pragma omp atomic
                        a[5] += g_o;
                    }
                    for (int k = 0; k < D; ++k) buffer[k] += col[k] * fac;
                }
            }
    }
    for (int g = 0; g < V; ++g) {
        const double* a = acc + (size_t)g * S;
        v_means2d[2 * g] = (float)a[0]; v_means2d[2 * g + 1] = (float)a[1];
        v_conics[3 * g] = (float)a[2]; v_conics[3 * g + 1] = (float)a[3]; v_conics[3 * g + 2] = (float)a[4];
        v_opacities[g] = (float)a[5];
        for (int k = 0; k < D; ++k) v_colors[(size_t)g * D + k] = (float)a[6 + k];
    }
    free(acc);
}

/* ------------------------------------------------------------------------- */
/* A7: projection backward (gsplat fully_fused_projection_packed_bwd) + the
 * gather backward of A1'.  Inputs are per-visible grads; outputs are DENSE
 * [N,*] (sparse_grad=False at rfstudio/model/gsplat.py:349), zero for culled
 * Gaussians.  v_opacities_packed / v_colors_packed are grads w.r.t. the
 * packed (opacity*compensation) and packed colors.                           */
GSO_API void gso_project_bwd(int N, int V, int D,
                             const float* means, const float* quats, const float* scales, const float* opacities,
                             const float* viewmat, const float* K, int W, int H, float eps2d,
                             const int32_t* gaussian_ids, const float* conics, const float* compensations,
                             const float* v_means2d, const float* v_depths /*nullable*/, const float* v_conics,
                             const float* v_opacities_packed, const float* v_colors_packed,
                             float* v_means, float* v_quats, float* v_scales, float* v_opacities, float* v_colors)
{
    gso_cam cam; cam_from(viewmat, K, W, H, &cam);
    memset(v_means, 0, sizeof(float) * 3 * (size_t)N);
    memset(v_quats, 0, sizeof(float) * 4 * (size_t)N);
    memset(v_scales, 0, sizeof(float) * 3 * (size_t)N);
    memset(v_opacities, 0, sizeof(float) * (size_t)N);
    memset(v_colors, 0, sizeof(float) * (size_t)D * (size_t)N);
#pragma omp parallel for schedule(static)
    for (int v = 0; v < V; ++v) {
        int n = gaussian_ids[v];
        const float* mean = means + 3 * (size_t)n;
        const float* quat = quats + 4 * (size_t)n;
        const float* scale = scales + 3 * (size_t)n;
        /* A1' backward: opac_packed = opac[n] * comp ; colors_packed = colors[n] */
        float comp = compensations[v];
        float v_op = v_opacities_packed[v];
        v_opacities[n] = v_op * comp;
        float v_comp = v_op * opacities[n];
        for (int k = 0; k < D; ++k) v_colors[(size_t)n * D + k] = v_colors_packed[(size_t)v * D + k];

        /* recompute forward intermediates */
        gso_proj p;
        project_one(&cam, mean, quat, scale, eps2d, -INFINITY, INFINITY, -1.0f, &p);
        const float* Cc = p.covar_c;

        /* conic = inverse(covar2d_blur): v_covar2d = -conic * v_conic_mat * conic */
        float ia = conics[3 * v], ib = conics[3 * v + 1], ic = conics[3 * v + 2];
        float ga = v_conics[3 * v], gb = 0.5f * v_conics[3 * v + 1], gc = v_conics[3 * v + 2];
        /* P = Minv * G */
        float p00 = ia * ga + ib * gb, p01 = ia * gb + ib * gc;
        float p10 = ib * ga + ic * gb, p11 = ib * gb + ic * gc;
        /* v_cov = -(P * Minv) */
        float vc00 = -(p00 * ia + p01 * ib), vc01 = -(p00 * ib + p01 * ic);
        float vc10 = -(p10 * ia + p11 * ib), vc11 = -(p10 * ib + p11 * ic);

        /* compensation vjp (gsplat add_blur_vjp) */
        {
            float det_conic = ia * ic - ib * ib;
            float v_sqr_comp = v_comp * 0.5f / (comp + 1e-6f);
            float one_minus_sqr_comp = 1.0f - comp * comp;
            vc00 += v_sqr_comp * (one_minus_sqr_comp * ia - eps2d * det_conic);
            vc01 += v_sqr_comp * (one_minus_sqr_comp * ib);
            vc10 += v_sqr_comp * (one_minus_sqr_comp * ib);
            vc11 += v_sqr_comp * (one_minus_sqr_comp * ic - eps2d * det_conic);
        }

        /* persp_proj vjp */
        float x = p.mean_c[0], y = p.mean_c[1], z = p.mean_c[2];
        float Wf = (float)W, Hf = (float)H;
        float tan_fovx = 0.5f * Wf / cam.fx, tan_fovy = 0.5f * Hf / cam.fy;
        float lim_x_pos = (Wf - cam.cx) / cam.fx + 0.3f * tan_fovx;
        float lim_x_neg = cam.cx / cam.fx + 0.3f * tan_fovx;
        float lim_y_pos = (Hf - cam.cy) / cam.fy + 0.3f * tan_fovy;
        float lim_y_neg = cam.cy / cam.fy + 0.3f * tan_fovy;
        float rz = 1.0f / z, rz2 = rz * rz, rz3 = rz2 * rz;
        float tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, x * rz));
        float ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, y * rz));
        float J[6] = { cam.fx * rz, 0.0f, -cam.fx * tx * rz2,
                       0.0f, cam.fy * rz, -cam.fy * ty * rz2 };       /* 2x3 row-major */
        float G[4] = { vc00, vc01, vc10, vc11 };
        /* v_covar_c = J^T G J */
        float GJ[6];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) GJ[i * 3 + j] = G[i * 2 + 0] * J[0 * 3 + j] + G[i * 2 + 1] * J[1 * 3 + j];
        float v_Cc[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) v_Cc[i * 3 + j] = J[0 * 3 + i] * GJ[0 * 3 + j] + J[1 * 3 + i] * GJ[1 * 3 + j];
        /* v_J = G J Cc^T + G^T J Cc */
        float JC[6], JCt[6];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) {
                JC[i * 3 + j]  = (J[i * 3 + 0] * Cc[0 * 3 + j] + J[i * 3 + 1] * Cc[1 * 3 + j]) + J[i * 3 + 2] * Cc[2 * 3 + j];
                JCt[i * 3 + j] = (J[i * 3 + 0] * Cc[j * 3 + 0] + J[i * 3 + 1] * Cc[j * 3 + 1]) + J[i * 3 + 2] * Cc[j * 3 + 2];
            }
        float v_J[6];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j)
                v_J[i * 3 + j] = (G[i * 2 + 0] * JCt[0 * 3 + j] + G[i * 2 + 1] * JCt[1 * 3 + j])
                               + (G[0 * 2 + i] * JC[0 * 3 + j] + G[1 * 2 + i] * JC[1 * 3 + j]);
        float vm2x = v_means2d[2 * v], vm2y = v_means2d[2 * v + 1];
        float v_mc[3];
        v_mc[0] = cam.fx * rz * vm2x;
        v_mc[1] = cam.fy * rz * vm2y;
        v_mc[2] = -(cam.fx * x * vm2x + cam.fy * y * vm2y) * rz2;
        if (x * rz <= lim_x_pos && x * rz >= -lim_x_neg) v_mc[0] += -cam.fx * rz2 * v_J[2];
        else                                             v_mc[2] += -cam.fx * rz3 * v_J[2] * tx;
        if (y * rz <= lim_y_pos && y * rz >= -lim_y_neg) v_mc[1] += -cam.fy * rz2 * v_J[5];
        else                                             v_mc[2] += -cam.fy * rz3 * v_J[5] * ty;
        v_mc[2] += -cam.fx * rz2 * v_J[0] - cam.fy * rz2 * v_J[4]
                 + 2.0f * cam.fx * tx * rz3 * v_J[2] + 2.0f * cam.fy * ty * rz3 * v_J[5];
        if (v_depths) v_mc[2] += v_depths[v];

        /* world<-camera: mean_c = R p + t ; covar_c = R covar R^T */
        const float* R = cam.R;
        for (int i = 0; i < 3; ++i)
            v_means[3 * (size_t)n + i] = (R[0 * 3 + i] * v_mc[0] + R[1 * 3 + i] * v_mc[1]) + R[2 * 3 + i] * v_mc[2];
        float Tm[9], v_cov[9];
        /* v_covar = R^T v_Cc R */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                Tm[i * 3 + j] = (R[0 * 3 + i] * v_Cc[0 * 3 + j] + R[1 * 3 + i] * v_Cc[1 * 3 + j]) + R[2 * 3 + i] * v_Cc[2 * 3 + j];
        mat3_mul(Tm, R, v_cov);

        /* covar = M M^T, M = Rq diag(s):  v_M = (v_cov + v_cov^T) M */
        const float* Rq = p.Rq;
        float M[9], v_M[9], Sy[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { M[i * 3 + j] = Rq[i * 3 + j] * scale[j]; Sy[i * 3 + j] = v_cov[i * 3 + j] + v_cov[j * 3 + i]; }
        mat3_mul(Sy, M, v_M);
        float v_Rq[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) v_Rq[i * 3 + j] = v_M[i * 3 + j] * scale[j];
        for (int j = 0; j < 3; ++j)
            v_scales[3 * (size_t)n + j] = (Rq[0 * 3 + j] * v_M[0 * 3 + j] + Rq[1 * 3 + j] * v_M[1 * 3 + j]) + Rq[2 * 3 + j] * v_M[2 * 3 + j];

        /* rotmat <- normalized quaternion <- raw quaternion */
        float qw = quat[0], qx = quat[1], qy = quat[2], qz = quat[3];
        float inv = 1.0f / sqrtf(((qx * qx + qy * qy) + qz * qz) + qw * qw);
        float w = qw * inv, xq = qx * inv, yq = qy * inv, zq = qz * inv;
#define VR(i, j) v_Rq[(i) * 3 + (j)]
        float vqn[4];
        vqn[0] = 2.0f * (xq * (VR(2, 1) - VR(1, 2)) + yq * (VR(0, 2) - VR(2, 0)) + zq * (VR(1, 0) - VR(0, 1)));
        vqn[1] = 2.0f * (-2.0f * xq * (VR(1, 1) + VR(2, 2)) + yq * (VR(1, 0) + VR(0, 1)) + zq * (VR(2, 0) + VR(0, 2)) + w * (VR(2, 1) - VR(1, 2)));
        vqn[2] = 2.0f * (xq * (VR(1, 0) + VR(0, 1)) - 2.0f * yq * (VR(0, 0) + VR(2, 2)) + zq * (VR(2, 1) + VR(1, 2)) + w * (VR(0, 2) - VR(2, 0)));
        vqn[3] = 2.0f * (xq * (VR(2, 0) + VR(0, 2)) + yq * (VR(2, 1) + VR(1, 2)) - 2.0f * zq * (VR(0, 0) + VR(1, 1)) + w * (VR(1, 0) - VR(0, 1)));
#undef VR
        float qn[4] = { w, xq, yq, zq };
        float dotp = ((vqn[0] * qn[0] + vqn[1] * qn[1]) + vqn[2] * qn[2]) + vqn[3] * qn[3];
        for (int k = 0; k < 4; ++k) v_quats[4 * (size_t)n + k] = (vqn[k] - dotp * qn[k]) * inv;
    }
}

/* ------------------------------------------------------------------------- */
/* S4: tone mapping (rfstudio/model/geosplat.py:474-480).
 * naive: rgb' = 1 - softplus_{beta=100, threshold=20}(1 - rgb*exposure); alpha passthrough.
 * mode 0 = none, 1 = naive, 2 = aces.  'none' is `render_rgba * exposure` (geosplat.py:123-124): the product covers
 * all FOUR channels, alpha included.                                          */
static inline float softplus100(float x)
{
    /* torch.nn.Softplus(beta=100, threshold=20): x if beta*x > 20 else log1p(exp(beta*x))/beta */
    float bx = 100.0f * x;
    return bx > 20.0f ? x : log1pf(expf(bx)) / 100.0f;
}
GSO_API void gso_tonemap_fwd(int64_t P, int mode, const float* rgba, float exposure, float* out)
{
    for (int64_t p = 0; p < P; ++p) {
        for (int k = 0; k < 3; ++k) {
            float rgb = rgba[4 * p + k] * exposure;
            float o;
            if (mode == 1) o = 1.0f - softplus100(1.0f - rgb);
            else if (mode == 2) o = (rgb * (2.51f * rgb + 0.03f)) / (rgb * (2.43f * rgb + 0.59f) + 0.14f);
            else o = rgb;
            out[4 * p + k] = o;
        }
        out[4 * p + 3] = mode == 0 ? rgba[4 * p + 3] * exposure : rgba[4 * p + 3];
    }
}
/* returns d/d(exposure) in *v_exposure (double-accumulated), writes v_rgba */
GSO_API void gso_tonemap_bwd(int64_t P, int mode, const float* rgba, float exposure, const float* v_out,
                             float* v_rgba, float* v_exposure)
{
    double ve = 0.0;
    for (int64_t p = 0; p < P; ++p) {
        for (int k = 0; k < 3; ++k) {
            float c = rgba[4 * p + k];
            float rgb = c * exposure;
            float d;
            if (mode == 1) {
                float bx = 100.0f * (1.0f - rgb);
                /* d/dx softplus = sigmoid(beta x) below threshold, 1 above; out = 1 - sp(1-rgb) -> d out/d rgb = sp' */
                d = bx > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-bx));
            } else if (mode == 2) {
                float num = rgb * (2.51f * rgb + 0.03f), den = rgb * (2.43f * rgb + 0.59f) + 0.14f;
                d = ((5.02f * rgb + 0.03f) * den - num * (4.86f * rgb + 0.59f)) / (den * den);
            } else d = 1.0f;
            float g = v_out[4 * p + k] * d;
            v_rgba[4 * p + k] = g * exposure;
            ve += (double)(g * c);
        }
        if (mode == 0) {
            v_rgba[4 * p + 3] = v_out[4 * p + 3] * exposure;
            ve += (double)(v_out[4 * p + 3] * rgba[4 * p + 3]);
        } else
            v_rgba[4 * p + 3] = v_out[4 * p + 3];
    }
    *v_exposure = (float)ve;
}
