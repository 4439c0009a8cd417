/*
 * gs_oracle_shade.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the per-Gaussian split-sum shading of the
 * GeoSplatting hot path (stages S1..S3 of SURVEY.md section 8a):
 *   S1  RenderableAttrs.splat arithmetic  rfstudio/model/geosplat.py:80-122
 *   S2  FG-LUT lookup (nvdiffrast dr.texture, 2-D, 'linear', 'clamp')
 *                                         rfstudio/model/geosplat.py:93-98
 *   S3  TextureSplitSum.sample            rfstudio/graphics/_mesh/_texture.py:571-613
 *       (dr.texture cube 'linear' on `base`, cube 'linear-mipmap-linear'
 *        on the specular pyramid with mip_level_bias = f(roughness))
 *
 * PARITY UNPINNED for the texture fetches: nvdiffrast is an un-vendored,
 * unpinned third-party dependency (README.md:36) and the reference holds no
 * golden vectors for it.  The S1 arithmetic and the roughness->mip map are
 * pinned against the importable reference Python (tests/golden/, generated
 * by scripts/make_golden.py).  Cube-map semantics restated here:
 *   - face order +x,-x,+y,-y,+z,-z with the (x,y) parameterisation of
 *     _cube_to_dir (rfstudio/graphics/_mesh/_texture.py:178-197) ==
 *     cube_to_dir (.../_splitsum/c_src/cubemap.cu:32-46); texel (ix,iy)
 *     centre at x = 2(ix+0.5)/R - 1;
 *   - bilinear footprint; a texel that falls over ONE face edge is taken
 *     from the adjacent face (re-projection of its centre, nearest texel);
 *     a texel over TWO edges (cube corner) does not exist: it takes the mean
 *     of the other three (the GL seamless-cube-map rule);
 *   - trilinear between floor(level) and floor(level)+1 with
 *     level = clamp(mip_level_bias, 0, L-1) (no uv derivatives are supplied
 *     by the reference, so the bias alone selects the level).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GSO_API __attribute__((visibility("default")))

/* ---------------- 2-D bilinear, clamp (S2) ------------------------------- */
/* lut layout [H][W][C]; u -> width axis, v -> height axis. */
static void tex2d_linear_clamp(const float* lut, int W, int H, int C, float u, float v,
                               float* out, float* d_du, float* d_dv)
{
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    int cx = 0, cy = 0;
    if (x < 0.0f) { x = 0.0f; cx = 1; } else if (x > (float)(W - 1)) { x = (float)(W - 1); cx = 1; }
    if (y < 0.0f) { y = 0.0f; cy = 1; } else if (y > (float)(H - 1)) { y = (float)(H - 1); cy = 1; }
    int ix0 = (int)floorf(x), iy0 = (int)floorf(y);
    float fx = x - (float)ix0, fy = y - (float)iy0;
    int ix1 = ix0 + 1 < W ? ix0 + 1 : W - 1;
    int iy1 = iy0 + 1 < H ? iy0 + 1 : H - 1;
    for (int c = 0; c < C; ++c) {
        float t00 = lut[((size_t)iy0 * W + ix0) * C + c], t10 = lut[((size_t)iy0 * W + ix1) * C + c];
        float t01 = lut[((size_t)iy1 * W + ix0) * C + c], t11 = lut[((size_t)iy1 * W + ix1) * C + c];
        float top = t00 + fx * (t10 - t00);
        float bot = t01 + fx * (t11 - t01);
        out[c] = top + fy * (bot - top);
        if (d_du) d_du[c] = cx ? 0.0f : (float)W * ((t10 - t00) + fy * ((t11 - t01) - (t10 - t00)));
        if (d_dv) d_dv[c] = cy ? 0.0f : (float)H * (bot - top);
    }
}

/* ---------------- cube map (S3) ------------------------------------------ */
typedef struct {
    int face;
    int a, b, c;        /* component indices: x_ndc = sx*d[a]/|d[c]|, y_ndc = sy*d[b]/|d[c]| */
    float sx, sy;
} face_map;

static const face_map FACES[6] = {
    { 0, 2, 1, 0, -1.0f, -1.0f },   /* +x: (1,-y,-x) */
    { 1, 2, 1, 0,  1.0f, -1.0f },   /* -x: (-1,-y,x) */
    { 2, 0, 2, 1,  1.0f,  1.0f },   /* +y: (x,1,y)   */
    { 3, 0, 2, 1,  1.0f, -1.0f },   /* -y: (x,-1,-y) */
    { 4, 0, 1, 2,  1.0f, -1.0f },   /* +z: (x,-y,1)  */
    { 5, 0, 1, 2, -1.0f, -1.0f },   /* -z: (-x,-y,-1)*/
};

static int select_face(const float* d)
{
    float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]);
    int f;
    float c;
    if (az > fmaxf(ax, ay)) { f = 4; c = d[2]; }
    else if (ay > ax)       { f = 2; c = d[1]; }
    else                    { f = 0; c = d[0]; }
    if (c < 0.0f) f += 1;
    return f;
}

static void face_point(int s, float x, float y, float* p)
{
    switch (s) {
    case 0: p[0] = 1.0f;  p[1] = -y;    p[2] = -x;    break;
    case 1: p[0] = -1.0f; p[1] = -y;    p[2] = x;     break;
    case 2: p[0] = x;     p[1] = 1.0f;  p[2] = y;     break;
    case 3: p[0] = x;     p[1] = -1.0f; p[2] = -y;    break;
    case 4: p[0] = x;     p[1] = -y;    p[2] = 1.0f;  break;
    default: p[0] = -x;   p[1] = -y;    p[2] = -1.0f; break;
    }
}

/* resolve texel (ix,iy) of face s (each may be one step outside [0,R-1]);
 * returns linear index face*R*R + y*R + x, or -1 for a cube corner. */
static int resolve_texel(int s, int ix, int iy, int R)
{
    int ox = (ix < 0 || ix >= R), oy = (iy < 0 || iy >= R);
    if (!ox && !oy) return (s * R + iy) * R + ix;
    if (ox && oy) return -1;
    float xn = 2.0f * (((float)ix + 0.5f) / (float)R) - 1.0f;
    float yn = 2.0f * (((float)iy + 0.5f) / (float)R) - 1.0f;
    float p[3];
    face_point(s, xn, yn, p);
    int s2 = select_face(p);
    const face_map* m = &FACES[s2];
    float inv = 1.0f / fabsf(p[m->c]);
    float x2 = m->sx * p[m->a] * inv, y2 = m->sy * p[m->b] * inv;
    float tx = (x2 + 1.0f) * 0.5f * (float)R - 0.5f, ty = (y2 + 1.0f) * 0.5f * (float)R - 0.5f;
    int jx = (int)floorf(tx + 0.5f), jy = (int)floorf(ty + 0.5f);
    if (jx < 0) jx = 0; if (jx > R - 1) jx = R - 1;
    if (jy < 0) jy = 0; if (jy > R - 1) jy = R - 1;
    return (s2 * R + jy) * R + jx;
}

typedef struct {
    int valid;           /* 0 -> zero direction: output 0, no grads */
    int idx[4];          /* texel linear indices (00,10,01,11); -1 = corner-missing */
    float w[4];          /* effective weights (missing weight redistributed) */
    float fx, fy;
    int face;
    float inv_c;         /* 1/|d[c]| */
    float xn, yn;        /* face coords in [-1,1] */
} cube_fp;

static void cube_footprint(const float* d, int R, cube_fp* fp)
{
    int s = select_face(d);
    const face_map* m = &FACES[s];
    float ac = fabsf(d[m->c]);
    fp->valid = (ac > 0.0f) && isfinite(ac);
    fp->face = s;
    if (!fp->valid) return;
    float inv = 1.0f / ac;
    float xn = m->sx * d[m->a] * inv, yn = m->sy * d[m->b] * inv;
    fp->inv_c = inv; fp->xn = xn; fp->yn = yn;
    float tx = (xn + 1.0f) * 0.5f * (float)R - 0.5f, ty = (yn + 1.0f) * 0.5f * (float)R - 0.5f;
    int ix0 = (int)floorf(tx), iy0 = (int)floorf(ty);
    float fx = tx - (float)ix0, fy = ty - (float)iy0;
    fp->fx = fx; fp->fy = fy;
    fp->idx[0] = resolve_texel(s, ix0, iy0, R);
    fp->idx[1] = resolve_texel(s, ix0 + 1, iy0, R);
    fp->idx[2] = resolve_texel(s, ix0, iy0 + 1, R);
    fp->idx[3] = resolve_texel(s, ix0 + 1, iy0 + 1, R);
    float w[4] = { (1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy };
    int miss = -1;
    for (int i = 0; i < 4; ++i) if (fp->idx[i] < 0) miss = i;
    for (int i = 0; i < 4; ++i) fp->w[i] = w[i];
    if (miss >= 0) {
        float wm = w[miss] / 3.0f;
        for (int i = 0; i < 4; ++i) fp->w[i] = (i == miss) ? 0.0f : w[i] + wm;
    }
}

/* bilinear cube fetch of 3 channels; optionally d out / d direction (3x3: dd[c*3+k]) */
static void cube_fetch(const float* tex, int R, const float* d, float* out, float* dd /*nullable [9]*/, cube_fp* fp_out)
{
    cube_fp fp;
    cube_footprint(d, R, &fp);
    if (fp_out) *fp_out = fp;
    if (!fp.valid) {
        out[0] = out[1] = out[2] = 0.0f;
        if (dd) memset(dd, 0, sizeof(float) * 9);
        return;
    }
    float t[4][3];
    int miss = -1;
    for (int i = 0; i < 4; ++i) {
        if (fp.idx[i] < 0) { miss = i; continue; }
        for (int c = 0; c < 3; ++c) t[i][c] = tex[(size_t)fp.idx[i] * 3 + c];
    }
    if (miss >= 0)
        for (int c = 0; c < 3; ++c) {
            float s = 0.0f;
            for (int i = 0; i < 4; ++i) if (i != miss) s += t[i][c];
            t[miss][c] = s / 3.0f;
        }
    const face_map* m = &FACES[fp.face];
    float sgn_c = d[m->c] < 0.0f ? -1.0f : 1.0f;
    for (int c = 0; c < 3; ++c) {
        float top = t[0][c] + fp.fx * (t[1][c] - t[0][c]);
        float bot = t[2][c] + fp.fx * (t[3][c] - t[2][c]);
        out[c] = top + fp.fy * (bot - top);
        if (dd) {
            float dtx = (t[1][c] - t[0][c]) + fp.fy * ((t[3][c] - t[2][c]) - (t[1][c] - t[0][c]));
            float dty = bot - top;
            float gx = dtx * 0.5f * (float)R, gy = dty * 0.5f * (float)R;   /* d/dxn, d/dyn */
            float g[3] = { 0.0f, 0.0f, 0.0f };
            g[m->a] += gx * m->sx * fp.inv_c;
            g[m->b] += gy * m->sy * fp.inv_c;
            g[m->c] += -(gx * fp.xn + gy * fp.yn) * fp.inv_c * sgn_c;
            dd[c * 3 + 0] = g[0]; dd[c * 3 + 1] = g[1]; dd[c * 3 + 2] = g[2];
        }
    }
}

static void cube_scatter(float* grad_tex, const cube_fp* fp, const float* g /*[3]*/, float scale)
{
    if (!fp->valid) return;
    for (int i = 0; i < 4; ++i) {
        if (fp->idx[i] < 0) continue;
        for (int c = 0; c < 3; ++c) {
            float v = g[c] * scale * fp->w[i];
#pragma omp atomic
            grad_tex[(size_t)fp->idx[i] * 3 + c] += v;
        }
    }
}

/* roughness -> mip level (rfstudio/graphics/_mesh/_texture.py:584-594) */
static float mip_from_roughness(float r, float min_r, float max_r, int L, float* dmip_dr)
{
    float m, dm;
    if (r < max_r) {
        float t = (r - min_r) / (max_r - min_r);
        int inside = (t >= 0.0f && t <= 1.0f);
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        m = t * (float)(L - 2);
        dm = inside ? (float)(L - 2) / (max_r - min_r) : 0.0f;
    } else {
        float t = (r - max_r) / (1.0f - max_r);
        int inside = (t >= 0.0f && t <= 1.0f);
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        m = t + (float)(L - 2);
        dm = inside ? 1.0f / (1.0f - max_r) : 0.0f;
    }
    if (dmip_dr) *dmip_dr = dm;
    return m;
}

typedef struct {
    float out[3];
    float dd[9];         /* d out_c / d dir_k */
    float dmip[3];       /* d out_c / d level */
    cube_fp fp0, fp1;
    float f;             /* blend weight of level l1 */
    int l0, l1;          /* l1 = -1 when single level */
} mip_sample;

static void cube_mip_fetch(const float* const* levels, const int* res, int L, const float* d, float bias, mip_sample* s)
{
    float lam = fminf(fmaxf(bias, 0.0f), (float)(L - 1));
    int l0 = (int)floorf(lam);
    if (l0 >= L - 1) {
        s->l0 = L - 1; s->l1 = -1; s->f = 0.0f;
        cube_fetch(levels[L - 1], res[L - 1], d, s->out, s->dd, &s->fp0);
        s->dmip[0] = s->dmip[1] = s->dmip[2] = 0.0f;
        return;
    }
    float f = lam - (float)l0;
    float c0[3], c1[3], dd0[9], dd1[9];
    cube_fetch(levels[l0], res[l0], d, c0, dd0, &s->fp0);
    cube_fetch(levels[l0 + 1], res[l0 + 1], d, c1, dd1, &s->fp1);
    s->l0 = l0; s->l1 = l0 + 1; s->f = f;
    int clamped = (bias < 0.0f || bias > (float)(L - 1));
    for (int c = 0; c < 3; ++c) {
        s->out[c] = c0[c] + f * (c1[c] - c0[c]);
        s->dmip[c] = clamped ? 0.0f : (c1[c] - c0[c]);
        for (int k = 0; k < 3; ++k) s->dd[c * 3 + k] = dd0[c * 3 + k] + f * (dd1[c * 3 + k] - dd0[c * 3 + k]);
    }
}

/* ---------------- S1: per-Gaussian shading ------------------------------- */
typedef struct {
    float rough, metal, spec[3], diff[3];
    float wo[3], len; int wo_const;
    float d, ndv;
    float fg[2], dfg_du[2], dfg_dv[2];
    float refl[3];
    float mip, dmip_dr;
    mip_sample ls;
    float ld[3], ld_dd[9]; cube_fp ld_fp;
    float refl_c[3];   /* reflectance */
} shade_tmp;

static void shade_one(const float* mean, const float* normal, const float* kd, const float* ks, const float* cam_pos,
                      float min_roughness, float max_metallic, int mode,
                      const float* lut, int lut_res, const float* base, int base_res,
                      const float* const* levels, const int* res, int L, float env_min_r, float env_max_r,
                      float* color, shade_tmp* t)
{
    t->rough = ks[0] * (1.0f - min_roughness) + min_roughness;
    t->metal = ks[1] * max_metallic;
    for (int c = 0; c < 3; ++c) {
        t->spec[c] = (1.0f - t->metal) * 0.04f + kd[c] * t->metal;
        t->diff[c] = kd[c] * (1.0f - t->metal);
    }
    /* safe_normalize (rfstudio/graphics/math.py:119-125) */
    float v[3] = { cam_pos[0] - mean[0], cam_pos[1] - mean[1], cam_pos[2] - mean[2] };
    float len = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    t->len = len;
    if (len < 1e-6f) { t->wo[0] = 0.0f; t->wo[1] = 0.0f; t->wo[2] = 1.0f; t->wo_const = 1; }
    else { float l = fmaxf(len, 1e-6f); t->wo[0] = v[0] / l; t->wo[1] = v[1] / l; t->wo[2] = v[2] / l; t->wo_const = 0; }
    t->d = (normal[0] * t->wo[0] + normal[1] * t->wo[1]) + normal[2] * t->wo[2];
    t->ndv = fmaxf(t->d, 1e-6f);
    tex2d_linear_clamp(lut, lut_res, lut_res, 2, t->ndv, t->rough, t->fg, t->dfg_du, t->dfg_dv);
    for (int k = 0; k < 3; ++k) t->refl[k] = 2.0f * t->d * normal[k] - t->wo[k];
    t->mip = mip_from_roughness(t->rough, env_min_r, env_max_r, L, &t->dmip_dr);
    cube_mip_fetch(levels, res, L, t->refl, t->mip, &t->ls);
    cube_fetch(base, base_res, normal, t->ld, t->ld_dd, &t->ld_fp);
    for (int c = 0; c < 3; ++c) {
        t->refl_c[c] = t->spec[c] * t->fg[0] + t->fg[1];
        if (mode == 0)      color[c] = t->diff[c] + t->ls.out[c] * t->refl_c[c];   /* 'pbr'      geosplat.py:111-115 */
        else if (mode == 1) color[c] = t->ld[c] * t->diff[c];                      /* 'diffuse'  geosplat.py:116-117 */
        else                color[c] = t->ls.out[c] * t->refl_c[c];                /* 'specular' geosplat.py:118-119 */
    }
}

GSO_API void gso_shade_fwd(int N, const float* means, const float* normals, const float* kd, const float* ks,
                           const float* cam_pos, float min_roughness, float max_metallic, int mode,
                           const float* lut, int lut_res, const float* base, int base_res,
                           const float* const* levels, const int* res, int L, float env_min_r, float env_max_r,
                           float* colors)
{
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        shade_tmp t;
        shade_one(means + 3 * (size_t)n, normals + 3 * (size_t)n, kd + 3 * (size_t)n, ks + 2 * (size_t)n, cam_pos,
                  min_roughness, max_metallic, mode, lut, lut_res, base, base_res, levels, res, L,
                  env_min_r, env_max_r, colors + 3 * (size_t)n, &t);
    }
}

/* v_base / v_levels[l] must be zero-initialised by the caller (accumulated into). */
GSO_API void gso_shade_bwd(int N, const float* means, const float* normals, const float* kd, const float* ks,
                           const float* cam_pos, float min_roughness, float max_metallic, int mode,
                           const float* lut, int lut_res, const float* base, int base_res,
                           const float* const* levels, const int* res, int L, float env_min_r, float env_max_r,
                           const float* v_colors,
                           float* v_means, float* v_normals, float* v_kd, float* v_ks,
                           float* v_base, float* const* v_levels)
{
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const float* normal = normals + 3 * (size_t)n;
        const float* kdn = kd + 3 * (size_t)n;
        const float* g = v_colors + 3 * (size_t)n;
        shade_tmp t;
        float color[3];
        shade_one(means + 3 * (size_t)n, normal, kdn, ks + 2 * (size_t)n, cam_pos,
                  min_roughness, max_metallic, mode, lut, lut_res, base, base_res, levels, res, L,
                  env_min_r, env_max_r, color, &t);
        float v_diff[3] = { 0, 0, 0 }, v_ls[3] = { 0, 0, 0 }, v_rf[3] = { 0, 0, 0 }, v_ld[3] = { 0, 0, 0 };
        for (int c = 0; c < 3; ++c) {
            if (mode == 0)      { v_diff[c] = g[c]; v_ls[c] = g[c] * t.refl_c[c]; v_rf[c] = g[c] * t.ls.out[c]; }
            else if (mode == 1) { v_ld[c] = g[c] * t.diff[c]; v_diff[c] = g[c] * t.ld[c]; }
            else                { v_ls[c] = g[c] * t.refl_c[c]; v_rf[c] = g[c] * t.ls.out[c]; }
        }
        float v_A = 0.0f, v_B = 0.0f, v_metal = 0.0f;
        for (int c = 0; c < 3; ++c) {
            float v_spec = v_rf[c] * t.fg[0];
            v_A += v_rf[c] * t.spec[c];
            v_B += v_rf[c];
            v_kd[3 * (size_t)n + c] = v_spec * t.metal + v_diff[c] * (1.0f - t.metal);
            v_metal += v_spec * (kdn[c] - 0.04f) - v_diff[c] * kdn[c];
        }
        float v_ndv = v_A * t.dfg_du[0] + v_B * t.dfg_du[1];
        float v_rough = v_A * t.dfg_dv[0] + v_B * t.dfg_dv[1];
        float v_mip = 0.0f, v_refl[3] = { 0, 0, 0 }, v_n[3] = { 0, 0, 0 };
        for (int c = 0; c < 3; ++c) {
            v_mip += v_ls[c] * t.ls.dmip[c];
            for (int k = 0; k < 3; ++k) {
                v_refl[k] += v_ls[c] * t.ls.dd[c * 3 + k];
                v_n[k] += v_ld[c] * t.ld_dd[c * 3 + k];
            }
        }
        v_rough += v_mip * t.dmip_dr;
        v_ks[2 * (size_t)n + 0] = v_rough * (1.0f - min_roughness);
        v_ks[2 * (size_t)n + 1] = v_metal * max_metallic;
        /* refl = 2 d n - wo */
        float v_d = 2.0f * ((v_refl[0] * normal[0] + v_refl[1] * normal[1]) + v_refl[2] * normal[2]);
        float v_wo[3];
        for (int k = 0; k < 3; ++k) { v_n[k] += 2.0f * t.d * v_refl[k]; v_wo[k] = -v_refl[k]; }
        if (t.d >= 1e-6f) v_d += v_ndv;
        for (int k = 0; k < 3; ++k) { v_n[k] += v_d * t.wo[k]; v_wo[k] += v_d * normal[k]; }
        for (int k = 0; k < 3; ++k) v_normals[3 * (size_t)n + k] = v_n[k];
        if (t.wo_const) {
            for (int k = 0; k < 3; ++k) v_means[3 * (size_t)n + k] = 0.0f;
        } else {
            float dot = (t.wo[0] * v_wo[0] + t.wo[1] * v_wo[1]) + t.wo[2] * v_wo[2];
            float l = fmaxf(t.len, 1e-6f);
            for (int k = 0; k < 3; ++k) v_means[3 * (size_t)n + k] = -((v_wo[k] - t.wo[k] * dot) / l);
        }
        /* texel gradients */
        if (mode != 1) {
            if (t.ls.l1 < 0) cube_scatter(v_levels[t.ls.l0], &t.ls.fp0, v_ls, 1.0f);
            else {
                cube_scatter(v_levels[t.ls.l0], &t.ls.fp0, v_ls, 1.0f - t.ls.f);
                cube_scatter(v_levels[t.ls.l1], &t.ls.fp1, v_ls, t.ls.f);
            }
        } else {
            cube_scatter(v_base, &t.ld_fp, v_ld, 1.0f);
        }
    }
}

/* stand-alone texture entry points so that tests can probe S2/S3 directly */
GSO_API void gso_tex2d_linear_clamp(int n, const float* lut, int W, int H, int C, const float* uv, float* out,
                                    float* d_du, float* d_dv)
{
    for (int i = 0; i < n; ++i)
        tex2d_linear_clamp(lut, W, H, C, uv[2 * i], uv[2 * i + 1], out + (size_t)i * C,
                           d_du ? d_du + (size_t)i * C : NULL, d_dv ? d_dv + (size_t)i * C : NULL);
}
GSO_API void gso_cube_linear(int n, const float* tex, int R, const float* dirs, float* out, float* dd)
{
    for (int i = 0; i < n; ++i) cube_fetch(tex, R, dirs + 3 * (size_t)i, out + 3 * (size_t)i, dd ? dd + 9 * (size_t)i : NULL, NULL);
}
GSO_API void gso_cube_mip_linear(int n, const float* const* levels, const int* res, int L, const float* dirs,
                                 const float* bias, float* out, float* dd, float* dmip)
{
    for (int i = 0; i < n; ++i) {
        mip_sample s;
        cube_mip_fetch(levels, res, L, dirs + 3 * (size_t)i, bias[i], &s);
        for (int c = 0; c < 3; ++c) out[3 * (size_t)i + c] = s.out[c];
        if (dd) memcpy(dd + 9 * (size_t)i, s.dd, sizeof(float) * 9);
        if (dmip) memcpy(dmip + 3 * (size_t)i, s.dmip, sizeof(float) * 3);
    }
}
GSO_API float gso_mip_from_roughness(float r, float min_r, float max_r, int L)
{
    return mip_from_roughness(r, min_r, max_r, L, NULL);
}
