/*
 * gs_oracle_splitsum.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the split-sum environment-map prefilter (stage S5 of
 * SURVEY.md section 8a), i.e. of the reference's in-repo CUDA kernels
 *   rfstudio/graphics/_mesh/_splitsum/c_src/cubemap.cu:17-46   pixel_area, cube_to_dir
 *   rfstudio/graphics/_mesh/_splitsum/c_src/cubemap.cu:110-168 DiffuseCubemapFwd/BwdKernel
 *   rfstudio/graphics/_mesh/_splitsum/c_src/cubemap.cu:174-244 ndfGGX, SpecularBoundsKernel
 *   rfstudio/graphics/_mesh/_splitsum/c_src/cubemap.cu:246-350 SpecularCubemapFwd/BwdKernel
 * and of the mip chain _CubeMapMip (rfstudio/graphics/_mesh/_texture.py:199-226).
 * The reference's own sources are CUDA (__global__/atomicAdd) and cannot be
 * compiled with gcc; no golden vectors exist for them in the reference
 * (tests/graphics/test_splitsum.py only asserts isfinite) -> PARITY UNPINNED,
 * the restatement follows the cited lines statement by statement.
 *
 * Gradient accumulators are double (the CUDA kernels use fp32 atomics in
 * arbitrary order).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GSO_API __attribute__((visibility("default")))

static float pixel_area(int x, int y, int N)                      /* cubemap.cu:17-30 */
{
    if (N > 1) {
        int H = N / 2;
        x = abs(x - H);
        y = abs(y - H);
        float dx = atanf((float)(x + 1) / (float)H) - atanf((float)x / (float)H);
        float dy = atanf((float)(y + 1) / (float)H) - atanf((float)y / (float)H);
        return dx * dy;
    }
    return 1.0f;
}

static void safe_normalize3(float* v)                             /* vec3f.h:90-94 */
{
    float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (l > 0.0f) { v[0] /= l; v[1] /= l; v[2] /= l; }
    else { v[0] = v[1] = v[2] = 0.0f; }
}

static void cube_to_dir(int x, int y, int side, int N, float* d)  /* cubemap.cu:32-46 */
{
    float fx = 2.0f * (((float)x + 0.5f) / (float)N) - 1.0f;
    float fy = 2.0f * (((float)y + 0.5f) / (float)N) - 1.0f;
    switch (side) {
    case 0: d[0] = 1.0f;  d[1] = -fy;   d[2] = -fx;   break;
    case 1: d[0] = -1.0f; d[1] = -fy;   d[2] = fx;    break;
    case 2: d[0] = fx;    d[1] = 1.0f;  d[2] = fy;    break;
    case 3: d[0] = fx;    d[1] = -1.0f; d[2] = -fy;   break;
    case 4: d[0] = fx;    d[1] = -fy;   d[2] = 1.0f;  break;
    default: d[0] = -fx;  d[1] = -fy;   d[2] = -1.0f; break;
    }
    safe_normalize3(d);
}

static inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* ---------------- mip chain: 2x2 average (_texture.py:201-206) ----------- */
GSO_API void gso_cubemap_mip_fwd(int R, int C, const float* in /*[6,R,R,C]*/, float* out /*[6,R/2,R/2,C]*/)
{
    int H = R / 2;
    for (int s = 0; s < 6; ++s)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < H; ++x)
                for (int c = 0; c < C; ++c) {
                    const float* p = in + (((size_t)s * R + 2 * y) * R + 2 * x) * C + c;
                    /* F.avg_pool2d: sum of the window (row-major) * 1/4 */
                    float sum = ((p[0] + p[C]) + p[(size_t)R * C]) + p[(size_t)R * C + C];
                    out[(((size_t)s * H + y) * H + x) * C + c] = sum * 0.25f;
                }
}

/* ---------------- diffuse irradiance (cubemap.cu:110-168) ---------------- */
GSO_API void gso_diffuse_cubemap_fwd(int R, const float* cubemap /*[6,R,R,3]*/, float* out /*[6,R,R,3]*/)
{
#pragma omp parallel for schedule(static)
    for (int o = 0; o < 6 * R * R; ++o) {
        int pz = o / (R * R), py = (o / R) % R, px = o % R;
        float N[3]; cube_to_dir(px, py, pz, R, N);
        float col[3] = { 0, 0, 0 };
        for (int s = 0; s < 6; ++s)
            for (int y = 0; y < R; ++y)
                for (int x = 0; x < R; ++x) {
                    float L[3]; cube_to_dir(x, y, s, R, L);
                    float costheta = fminf(fmaxf(dot3(N, L), 0.0f), 0.999f);
                    float w = costheta * pixel_area(x, y, R) / 3.141592f;
                    const float* t = cubemap + (((size_t)s * R + y) * R + x) * 3;
                    col[0] += t[0] * w; col[1] += t[1] * w; col[2] += t[2] * w;
                }
        out[(size_t)o * 3] = col[0]; out[(size_t)o * 3 + 1] = col[1]; out[(size_t)o * 3 + 2] = col[2];
    }
}

GSO_API void gso_diffuse_cubemap_bwd(int R, const float* v_out /*[6,R,R,3]*/, float* v_cubemap /*[6,R,R,3]*/)
{
    size_t n = (size_t)6 * R * R;
    double* acc = (double*)calloc(n * 3, sizeof(double));
    /* adjoint: loop over INPUT texels so that no atomics are needed */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < (int)n; ++i) {
        int s = i / (R * R), y = (i / R) % R, x = i % R;
        float L[3]; cube_to_dir(x, y, s, R, L);
        float pa = pixel_area(x, y, R);
        double a0 = 0, a1 = 0, a2 = 0;
        for (int o = 0; o < (int)n; ++o) {
            int pz = o / (R * R), py = (o / R) % R, px = o % R;
            float N[3]; cube_to_dir(px, py, pz, R, N);
            float costheta = fminf(fmaxf(dot3(N, L), 0.0f), 0.999f);
            float w = costheta * pa / 3.141592f;
            a0 += (double)(v_out[(size_t)o * 3] * w);
            a1 += (double)(v_out[(size_t)o * 3 + 1] * w);
            a2 += (double)(v_out[(size_t)o * 3 + 2] * w);
        }
        acc[(size_t)i * 3] = a0; acc[(size_t)i * 3 + 1] = a1; acc[(size_t)i * 3 + 2] = a2;
    }
    for (size_t i = 0; i < n * 3; ++i) v_cubemap[i] = (float)acc[i];
    free(acc);
}

/* ---------------- GGX specular prefilter (cubemap.cu:174-350) ------------ */
static inline float ndfGGX(float alphaSqr, float cosTheta)        /* cubemap.cu:174-179 */
{
    float c = fminf(fmaxf(cosTheta, 0.0f), 1.0f);
    float d = (c * alphaSqr - c) * c + 1.0f;
    /* M_PI is a double constant in the CUDA source: (d*d) is promoted and the quotient rounded once */
    return (float)((double)alphaSqr / ((double)(d * d) * M_PI));
}

/* bounds[6,R,R,24] stored as floats, like the reference (SpecularBoundsKernel, cubemap.cu:181-244) */
GSO_API void gso_specular_bounds(int R, float costheta_cutoff, float* bounds)
{
    const int TILE = 16;
#pragma omp parallel for schedule(dynamic, 64)
    for (int o = 0; o < 6 * R * R; ++o) {
        int pz = o / (R * R), py = (o / R) % R, px = o % R;
        float VNR[3]; cube_to_dir(px, py, pz, R, VNR);
        for (int s = 0; s < 6; ++s) {
            int min_x = R - 1, max_x = 0, min_y = R - 1, max_y = 0;
            for (int tx = 0; tx < (R + TILE - 1) / TILE; ++tx)
                for (int ty = 0; ty < (R + TILE - 1) / TILE; ++ty) {
                    int tsx = tx * TILE, tsy = ty * TILE;
                    int tex = (tx + 1) * TILE < R ? (tx + 1) * TILE : R;
                    int tey = (ty + 1) * TILE < R ? (ty + 1) * TILE : R;
                    float L0[3], L1[3], L2[3], L3[3];
                    cube_to_dir(tsx, tsy, s, R, L0); cube_to_dir(tex, tsy, s, R, L1);
                    cube_to_dir(tsx, tey, s, R, L2); cube_to_dir(tex, tey, s, R, L3);
                    float mn[3], mx[3];
                    for (int k = 0; k < 3; ++k) {
                        mn[k] = fminf(fminf(L0[k], L1[k]), fminf(L2[k], L3[k]));
                        mx[k] = fmaxf(fmaxf(L0[k], L1[k]), fmaxf(L2[k], L3[k]));
                    }
                    float maxdp = fmaxf(mn[0] * VNR[0], mx[0] * VNR[0]) + fmaxf(mn[1] * VNR[1], mx[1] * VNR[1])
                                + fmaxf(mn[2] * VNR[2], mx[2] * VNR[2]);
                    if (maxdp >= costheta_cutoff) {
                        for (int y = tsy; y < tey; ++y)
                            for (int x = tsx; x < tex; ++x) {
                                float L[3]; cube_to_dir(x, y, s, R, L);
                                if (dot3(L, VNR) >= costheta_cutoff) {
                                    if (x < min_x) min_x = x; if (x > max_x) max_x = x;
                                    if (y < min_y) min_y = y; if (y > max_y) max_y = y;
                                }
                            }
                    }
                }
            float* b = bounds + (size_t)o * 24 + s * 4;
            b[0] = (float)min_x; b[1] = (float)max_x; b[2] = (float)min_y; b[3] = (float)max_y;
        }
    }
}

/* out[6,R,R,4] = (sum rgb*w, sum w) exactly like SpecularCubemapFwdKernel */
GSO_API void gso_specular_cubemap_fwd(int R, const float* cubemap, const float* bounds, float roughness,
                                      float costheta_cutoff, float* out)
{
    float alpha = roughness * roughness;
    float alphaSqr = alpha * alpha;
#pragma omp parallel for schedule(dynamic, 64)
    for (int o = 0; o < 6 * R * R; ++o) {
        int pz = o / (R * R), py = (o / R) % R, px = o % R;
        float VNR[3]; cube_to_dir(px, py, pz, R, VNR);
        float wsum = 0.0f, col[3] = { 0, 0, 0 };
        for (int s = 0; s < 6; ++s) {
            const float* b = bounds + (size_t)o * 24 + s * 4;
            int xmin = (int)b[0], xmax = (int)b[1], ymin = (int)b[2], ymax = (int)b[3];
            if (xmin <= xmax)
                for (int y = ymin; y <= ymax; ++y)
                    for (int x = xmin; x <= xmax; ++x) {
                        float L[3]; cube_to_dir(x, y, s, R, L);
                        if (dot3(L, VNR) >= costheta_cutoff) {
                            float Hv[3] = { L[0] + VNR[0], L[1] + VNR[1], L[2] + VNR[2] };
                            safe_normalize3(Hv);
                            float wiDotN = fmaxf(dot3(L, VNR), 0.0f);
                            float VNRDotH = fmaxf(dot3(VNR, Hv), 0.0f);
                            float w = wiDotN * ndfGGX(alphaSqr, VNRDotH) * pixel_area(x, y, R) / 4.0f;
                            const float* t = cubemap + (((size_t)s * R + y) * R + x) * 3;
                            col[0] += t[0] * w; col[1] += t[1] * w; col[2] += t[2] * w;
                            wsum += w;
                        }
                    }
        }
        out[(size_t)o * 4] = col[0]; out[(size_t)o * 4 + 1] = col[1]; out[(size_t)o * 4 + 2] = col[2];
        out[(size_t)o * 4 + 3] = wsum;
    }
}

/* v_out[6,R,R,3] is the gradient w.r.t. the UN-normalised rgb sums (channels 0..2 of the forward output) */
GSO_API void gso_specular_cubemap_bwd(int R, const float* bounds, const float* v_out, float roughness,
                                      float costheta_cutoff, float* v_cubemap)
{
    float alpha = roughness * roughness;
    float alphaSqr = alpha * alpha;
    size_t n = (size_t)6 * R * R;
    double* acc = (double*)calloc(n * 3, sizeof(double));
#pragma omp parallel for schedule(dynamic, 64)
    for (int o = 0; o < (int)n; ++o) {
        int pz = o / (R * R), py = (o / R) % R, px = o % R;
        float VNR[3]; cube_to_dir(px, py, pz, R, VNR);
        const float* g = v_out + (size_t)o * 3;
        for (int s = 0; s < 6; ++s) {
            const float* b = bounds + (size_t)o * 24 + s * 4;
            int xmin = (int)b[0], xmax = (int)b[1], ymin = (int)b[2], ymax = (int)b[3];
            if (xmin <= xmax)
                for (int y = ymin; y <= ymax; ++y)
                    for (int x = xmin; x <= xmax; ++x) {
                        float L[3]; cube_to_dir(x, y, s, R, L);
                        if (dot3(L, VNR) >= costheta_cutoff) {
                            float Hv[3] = { L[0] + VNR[0], L[1] + VNR[1], L[2] + VNR[2] };
                            safe_normalize3(Hv);
                            float wiDotN = fmaxf(dot3(L, VNR), 0.0f);
                            float VNRDotH = fmaxf(dot3(VNR, Hv), 0.0f);
                            float w = wiDotN * ndfGGX(alphaSqr, VNRDotH) * pixel_area(x, y, R) / 4.0f;
                            double* a = acc + (((size_t)s * R + y) * R + x) * 3;
                            for (int c = 0; c < 3; ++c) {
                                double v = (double)(g[c] * w);
#pragma omp atomic
                                a[c] += v;
                            }
                        }
                    }
        }
    }
    for (size_t i = 0; i < n * 3; ++i) v_cubemap[i] = (float)acc[i];
    free(acc);
}

/* ---------------- subsets of output texels (full-size parity tests: R = 512 / 256 / 128) ----------------
 * The prefilter is independent per output texel, so a test can evaluate it on a random subset `sel[n_sel]` of the
 * 6*R*R texels (flat index (s*R + y)*R + x) in seconds.  Same statements as the full loops above; bounds / out /
 * v_out are COMPACT (row i belongs to texel sel[i]), v_cubemap is the full [6,R,R,3] gradient. */
static void bounds_of_texel(int R, float costheta_cutoff, int o, float* b24)
{
    const int TILE = 16;
    int pz = o / (R * R), py = (o / R) % R, px = o % R;
    float VNR[3]; cube_to_dir(px, py, pz, R, VNR);
    for (int s = 0; s < 6; ++s) {
        int min_x = R - 1, max_x = 0, min_y = R - 1, max_y = 0;
        for (int tx = 0; tx < (R + TILE - 1) / TILE; ++tx)
            for (int ty = 0; ty < (R + TILE - 1) / TILE; ++ty) {
                int tsx = tx * TILE, tsy = ty * TILE;
                int tex = (tx + 1) * TILE < R ? (tx + 1) * TILE : R;
                int tey = (ty + 1) * TILE < R ? (ty + 1) * TILE : R;
                float L0[3], L1[3], L2[3], L3[3];
                cube_to_dir(tsx, tsy, s, R, L0); cube_to_dir(tex, tsy, s, R, L1);
                cube_to_dir(tsx, tey, s, R, L2); cube_to_dir(tex, tey, s, R, L3);
                float mn[3], mx[3];
                for (int k = 0; k < 3; ++k) {
                    mn[k] = fminf(fminf(L0[k], L1[k]), fminf(L2[k], L3[k]));
                    mx[k] = fmaxf(fmaxf(L0[k], L1[k]), fmaxf(L2[k], L3[k]));
                }
                float maxdp = fmaxf(mn[0] * VNR[0], mx[0] * VNR[0]) + fmaxf(mn[1] * VNR[1], mx[1] * VNR[1])
                            + fmaxf(mn[2] * VNR[2], mx[2] * VNR[2]);
                if (maxdp >= costheta_cutoff) {
                    for (int y = tsy; y < tey; ++y)
                        for (int x = tsx; x < tex; ++x) {
                            float L[3]; cube_to_dir(x, y, s, R, L);
                            if (dot3(L, VNR) >= costheta_cutoff) {
                                if (x < min_x) min_x = x; if (x > max_x) max_x = x;
                                if (y < min_y) min_y = y; if (y > max_y) max_y = y;
                            }
                        }
                }
            }
        float* b = b24 + s * 4;
        b[0] = (float)min_x; b[1] = (float)max_x; b[2] = (float)min_y; b[3] = (float)max_y;
    }
}

GSO_API void gso_specular_bounds_subset(int R, float costheta_cutoff, int n_sel, const int* sel, float* bounds /*[n_sel,24]*/)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n_sel; ++i) bounds_of_texel(R, costheta_cutoff, sel[i], bounds + (size_t)i * 24);
}

GSO_API void gso_specular_cubemap_fwd_subset(int R, const float* cubemap, int n_sel, const int* sel, const float* bounds,
                                             float roughness, float costheta_cutoff, float* out /*[n_sel,4]*/)
{
    float alpha = roughness * roughness;
    float alphaSqr = alpha * alpha;
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n_sel; ++i) {
        int o = sel[i];
        int pz = o / (R * R), py = (o / R) % R, px = o % R;
        float VNR[3]; cube_to_dir(px, py, pz, R, VNR);
        float wsum = 0.0f, col[3] = { 0, 0, 0 };
        for (int s = 0; s < 6; ++s) {
            const float* b = bounds + (size_t)i * 24 + s * 4;
            int xmin = (int)b[0], xmax = (int)b[1], ymin = (int)b[2], ymax = (int)b[3];
            if (xmin <= xmax)
                for (int y = ymin; y <= ymax; ++y)
                    for (int x = xmin; x <= xmax; ++x) {
                        float L[3]; cube_to_dir(x, y, s, R, L);
                        if (dot3(L, VNR) >= costheta_cutoff) {
                            float Hv[3] = { L[0] + VNR[0], L[1] + VNR[1], L[2] + VNR[2] };
                            safe_normalize3(Hv);
                            float wiDotN = fmaxf(dot3(L, VNR), 0.0f);
                            float VNRDotH = fmaxf(dot3(VNR, Hv), 0.0f);
                            float w = wiDotN * ndfGGX(alphaSqr, VNRDotH) * pixel_area(x, y, R) / 4.0f;
                            const float* t = cubemap + (((size_t)s * R + y) * R + x) * 3;
                            col[0] += t[0] * w; col[1] += t[1] * w; col[2] += t[2] * w;
                            wsum += w;
                        }
                    }
        }
        out[(size_t)i * 4] = col[0]; out[(size_t)i * 4 + 1] = col[1]; out[(size_t)i * 4 + 2] = col[2];
        out[(size_t)i * 4 + 3] = wsum;
    }
}

GSO_API void gso_specular_cubemap_bwd_subset(int R, int n_sel, const int* sel, const float* bounds, const float* v_out /*[n_sel,3]*/,
                                             float roughness, float costheta_cutoff, float* v_cubemap /*[6,R,R,3]*/)
{
    float alpha = roughness * roughness;
    float alphaSqr = alpha * alpha;
    size_t n = (size_t)6 * R * R;
    double* acc = (double*)calloc(n * 3, sizeof(double));
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n_sel; ++i) {
        int o = sel[i];
        int pz = o / (R * R), py = (o / R) % R, px = o % R;
        float VNR[3]; cube_to_dir(px, py, pz, R, VNR);
        const float* g = v_out + (size_t)i * 3;
        for (int s = 0; s < 6; ++s) {
            const float* b = bounds + (size_t)i * 24 + s * 4;
            int xmin = (int)b[0], xmax = (int)b[1], ymin = (int)b[2], ymax = (int)b[3];
            if (xmin <= xmax)
                for (int y = ymin; y <= ymax; ++y)
                    for (int x = xmin; x <= xmax; ++x) {
                        float L[3]; cube_to_dir(x, y, s, R, L);
                        if (dot3(L, VNR) >= costheta_cutoff) {
                            float Hv[3] = { L[0] + VNR[0], L[1] + VNR[1], L[2] + VNR[2] };
                            safe_normalize3(Hv);
                            float wiDotN = fmaxf(dot3(L, VNR), 0.0f);
                            float VNRDotH = fmaxf(dot3(VNR, Hv), 0.0f);
                            float w = wiDotN * ndfGGX(alphaSqr, VNRDotH) * pixel_area(x, y, R) / 4.0f;
                            double* a = acc + (((size_t)s * R + y) * R + x) * 3;
                            for (int c = 0; c < 3; ++c) {
                                double v = (double)(g[c] * w);
#pragma omp atomic
                                a[c] += v;
                            }
                        }
                    }
        }
    }
    for (size_t i = 0; i < n * 3; ++i) v_cubemap[i] = (float)acc[i];
    free(acc);
}
