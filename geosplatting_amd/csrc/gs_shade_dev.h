// gs_shade_dev.h -- device-side pieces of the shading stage (S1-S3) shared by gs_shade.hip (the RenderableAttrs.splat call shape)
// and gs_front.hip (the engine's fused front / tail kernels): environment descriptors, FG-LUT lookup, roughness -> mip level,
// trilinear cube fetch, the shading arithmetic of ONE Gaussian, LDS texel scatter, and the host-side layout of the texel-gradient
// accumulators.  Reference: rfstudio/model/geosplat.py:80-122, rfstudio/graphics/_mesh/_texture.py:571-613.
// The including file must have floating-point contraction OFF (texel / LUT-cell selection is discontinuous in the coordinates).
#pragma once
#include "gs_common.h"
#include <stdlib.h>
#include <math.h>
#include "gs_cube.h"

struct EnvDev {
    const float* lut; int lut_res;
    const float* base; int base_res;
    int L;
    const float* levels[GS_MAX_LEVELS];
    int res[GS_MAX_LEVELS];
    float min_r, max_r;
};
struct EnvGradDev {
    float* base;
    float* levels[GS_MAX_LEVELS];
    // LDS privatisation of the small, heavily contended levels: float offset into the block's LDS
    // accumulator or -1 (accumulate straight into HBM with atomics)
    int lds_base;
    int lds_level[GS_MAX_LEVELS];
    int lds_floats;
    int stage_off;                 // float offset of the per-wave commit staging (640 floats per wave) or -1
    // per-XCD private accumulators for the levels that do not fit LDS (XCD-local atomics, reduced afterwards):
    // copy x of level l lives at priv + x * priv_stride + priv_level[l] (floats); priv == nullptr -> device atomics
    float* priv;
    long long priv_stride;
    long long priv_level[GS_MAX_LEVELS];
    long long priv_base;
};

__device__ __forceinline__ void tex2d_linear_clamp2(const float* __restrict__ lut, int W, int H, float u, float v,
                                                    float* out, float* d_du, float* d_dv)
{
    float x = u * (float)W - 0.5f, y = v * (float)H - 0.5f;
    bool cx = false, cy = false;
    if (x < 0.0f) { x = 0.0f; cx = true; } else if (x > (float)(W - 1)) { x = (float)(W - 1); cx = true; }
    if (y < 0.0f) { y = 0.0f; cy = true; } else if (y > (float)(H - 1)) { y = (float)(H - 1); cy = true; }
    const int ix0 = (int)floorf(x), iy0 = (int)floorf(y);
    const float fx = x - (float)ix0, fy = y - (float)iy0;
    const int ix1 = min(ix0 + 1, W - 1), iy1 = min(iy0 + 1, H - 1);
    const float2 t00 = *reinterpret_cast<const float2*>(lut + ((size_t)iy0 * W + ix0) * 2);
    const float2 t10 = *reinterpret_cast<const float2*>(lut + ((size_t)iy0 * W + ix1) * 2);
    const float2 t01 = *reinterpret_cast<const float2*>(lut + ((size_t)iy1 * W + ix0) * 2);
    const float2 t11 = *reinterpret_cast<const float2*>(lut + ((size_t)iy1 * W + ix1) * 2);
    const float a00[2] = { t00.x, t00.y }, a10[2] = { t10.x, t10.y }, a01[2] = { t01.x, t01.y }, a11[2] = { t11.x, t11.y };
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float top = a00[c] + fx * (a10[c] - a00[c]);
        const float bot = a01[c] + fx * (a11[c] - a01[c]);
        out[c] = top + fy * (bot - top);
        d_du[c] = cx ? 0.0f : (float)W * ((a10[c] - a00[c]) + fy * ((a11[c] - a01[c]) - (a10[c] - a00[c])));
        d_dv[c] = cy ? 0.0f : (float)H * (bot - top);
    }
}

__device__ __forceinline__ float mip_from_roughness(float r, float min_r, float max_r, int L, float& dm)
{
    float m;
    if (r < max_r) {
        float t = (r - min_r) / (max_r - min_r);
        const bool in = (t >= 0.0f && t <= 1.0f);
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        m = t * (float)(L - 2);
        dm = in ? (float)(L - 2) / (max_r - min_r) : 0.0f;
    } else {
        float t = (r - max_r) / (1.0f - max_r);
        const bool in = (t >= 0.0f && t <= 1.0f);
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        m = t + (float)(L - 2);
        dm = in ? 1.0f / (1.0f - max_r) : 0.0f;
    }
    return m;
}

struct MipSample {
    float out[3], dd[9], dmip[3];
    CubeFp fp0, fp1;
    float f;
    int l0, l1;
};

template <bool WITH_GRAD>
__device__ void cube_mip_fetch(const EnvDev& env, const float* d, float bias, MipSample& s)
{
    const int L = env.L;
    const float lam = fminf(fmaxf(bias, 0.0f), (float)(L - 1));
    // no early-out for the last level (a branch would serialise the texel loads of the two fetches): it samples level
    // L-1 twice with f = 0, which gives bit-identical results
    const int lf = (int)floorf(lam);
    const bool last = lf >= L - 1;
    const int l0 = last ? L - 1 : lf;
    const int l1 = last ? L - 1 : lf + 1;
    const float f = last ? 0.0f : lam - (float)l0;
    float c0[3], c1[3], dd0[9], dd1[9];
    cube_fetch<WITH_GRAD>(env.levels[l0], env.res[l0], d, c0, dd0, s.fp0);
    cube_fetch<WITH_GRAD>(env.levels[l1], env.res[l1], d, c1, dd1, s.fp1);
    s.l0 = l0; s.l1 = last ? -1 : l1; s.f = f;
    const bool clamped = last || (bias < 0.0f || bias > (float)(L - 1));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        s.out[c] = c0[c] + f * (c1[c] - c0[c]);
        s.dmip[c] = clamped ? 0.0f : (c1[c] - c0[c]);
        if (WITH_GRAD) {
#pragma unroll
            for (int k = 0; k < 3; ++k) s.dd[c * 3 + k] = dd0[c * 3 + k] + f * (dd1[c * 3 + k] - dd0[c * 3 + k]);
        }
    }
}

struct ShadeTmp {
    float rough, metal, spec[3], diff[3];
    float wo[3], len; bool wo_const;
    float d, ndv, fg[2], dfg_du[2], dfg_dv[2];
    float refl[3], mip, dmip_dr;
    MipSample ls;
    float ld[3], ld_dd[9]; CubeFp ld_fp;
    float refl_c[3];
};

template <bool WITH_GRAD>
__device__ void shade_one(const float* mean, const float* normal, const float* kd, const float* ks, const float* cam_pos,
                          float min_roughness, float max_metallic, int mode, const EnvDev& env, float* color, ShadeTmp& t)
{
    t.rough = ks[0] * (1.0f - min_roughness) + min_roughness;
    t.metal = ks[1] * max_metallic;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t.spec[c] = (1.0f - t.metal) * 0.04f + kd[c] * t.metal;
        t.diff[c] = kd[c] * (1.0f - t.metal);
    }
    const float v[3] = { cam_pos[0] - mean[0], cam_pos[1] - mean[1], cam_pos[2] - mean[2] };
    const float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    t.len = len;
    if (len < 1e-6f) { t.wo[0] = 0.0f; t.wo[1] = 0.0f; t.wo[2] = 1.0f; t.wo_const = true; }
    else { const float l = fmaxf(len, 1e-6f); t.wo[0] = v[0] / l; t.wo[1] = v[1] / l; t.wo[2] = v[2] / l; t.wo_const = false; }
    t.d = normal[0] * t.wo[0] + normal[1] * t.wo[1] + normal[2] * t.wo[2];
    t.ndv = fmaxf(t.d, 1e-6f);
    tex2d_linear_clamp2(env.lut, env.lut_res, env.lut_res, t.ndv, t.rough, t.fg, t.dfg_du, t.dfg_dv);
#pragma unroll
    for (int k = 0; k < 3; ++k) t.refl[k] = 2.0f * t.d * normal[k] - t.wo[k];
    t.mip = mip_from_roughness(t.rough, env.min_r, env.max_r, env.L, t.dmip_dr);
    if (mode != GS_MODE_DIFFUSE) cube_mip_fetch<WITH_GRAD>(env, t.refl, t.mip, t.ls);
    else { t.ls.out[0] = t.ls.out[1] = t.ls.out[2] = 0.0f; }
    if (mode == GS_MODE_DIFFUSE) cube_fetch<WITH_GRAD>(env.base, env.base_res, normal, t.ld, t.ld_dd, t.ld_fp);
    else { t.ld[0] = t.ld[1] = t.ld[2] = 0.0f; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t.refl_c[c] = t.spec[c] * t.fg[0] + t.fg[1];
        if (mode == GS_MODE_PBR)          color[c] = t.diff[c] + t.ls.out[c] * t.refl_c[c];
        else if (mode == GS_MODE_DIFFUSE) color[c] = t.ld[c] * t.diff[c];
        else                              color[c] = t.ls.out[c] * t.refl_c[c];
    }
}

// ---- lean backward of one pair (round 4, tail_shade_pairs_kernel) -------------------------------------------------------------------
// shade_one<true> and the chain rule behind it carry the full 3x3 Jacobians d colour / d direction of both pyramid levels (113 VGPRs before the
// texel scatter).  The cotangent of the sampled colour is known BEFORE the cube fetch (v_ls = g * (spec * fg0 + fg1)), so the
// fetch can contract it on the spot: per level 4 scalars <v, tap> instead of 12 taps and 9 Jacobian entries.  Everything that
// SELECTS something (LUT cell, mip level, face, texel) is computed exactly as in the forward (contraction off); only the smooth
// arithmetic behind it is contracted -- gradients are compared to 1e-4, not bit for bit.
// bilinear cube sample, s = <v, sample> and v_d = d s / d direction
__device__ __forceinline__ void cube_fetch_vjp(const float* __restrict__ tex, int R, const float* d, const float* v, float* out,
                                               float& s, float* v_d, CubeFp& fp)
{
    cube_footprint(d, R, fp);
    const bool valid = fp.valid;
    bool ok[4]; int safe[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ok[i] = valid && fp.idx[i] >= 0; safe[i] = ok[i] ? fp.idx[i] : 0; }
    float t[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                   // branch-free loads (see cube_fetch)
        const float* p = tex + (size_t)safe[i] * 3;
        t[i][0] = p[0]; t[i][1] = p[1]; t[i][2] = p[2];
    }
    bool has_miss = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!ok[i]) { t[i][0] = t[i][1] = t[i][2] = 0.0f; }
        has_miss = has_miss || (valid && fp.idx[i] < 0);
    }
    if (has_miss) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float m = (t[0][c] + t[1][c] + t[2][c] + t[3][c]) / 3.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (fp.idx[i] < 0) t[i][c] = m;
        }
    }
    {
#pragma clang fp contract(fast)
        float q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = v[0] * t[i][0] + v[1] * t[i][1] + v[2] * t[i][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = t[0][c] + fp.fx * (t[1][c] - t[0][c]);
            const float bot = t[2][c] + fp.fx * (t[3][c] - t[2][c]);
            out[c] = valid ? top + fp.fy * (bot - top) : 0.0f;
        }
        const float qt = q[0] + fp.fx * (q[1] - q[0]), qb = q[2] + fp.fx * (q[3] - q[2]);
        s = valid ? qt + fp.fy * (qb - qt) : 0.0f;
        const float dtx = (q[1] - q[0]) + fp.fy * ((q[3] - q[2]) - (q[1] - q[0]));
        const float dty = qb - qt;
        const FaceMap m = face_map(fp.face);
        const float sgn_c = comp3(d, m.c) < 0.0f ? -1.0f : 1.0f;
        const float gx = dtx * 0.5f * (float)R, gy = dty * 0.5f * (float)R;
        const float va = gx * m.sx * fp.inv_c, vb = gy * m.sy * fp.inv_c;
        const float vc = -(gx * fp.xn + gy * fp.yn) * fp.inv_c * sgn_c;
#pragma unroll
        for (int k = 0; k < 3; ++k) v_d[k] = valid ? (k == m.a ? va : 0.0f) + (k == m.b ? vb : 0.0f) + (k == m.c ? vc : 0.0f) : 0.0f;
    }
}

// The pair's backward in three phases, so that a kernel can take the pyramid levels ONE AT A TIME (fetch + contract + scatter a
// level, then the next: the two fetches interleaved by the compiler cost 79 VGPRs, one costs 51):
//   shade_pair_pre  : everything before the cube fetch -- selecting quantities, FG LUT, the colour cotangent p.v, levels / weights
//   cube_fetch_vjp  : per level, on p.dir with cotangent p.v; the caller accumulates w * out, w * v_dir and (s1 - s0)
//   shade_pair_post : the remaining chain rule, ADDS into a_mean / a_n / a_kd / a_ks
struct PairPre {
    float metal, spec[3], wo[3], len, d;
    float fg0, dfg_du[2], dfg_dv[2];
    float dir[3];                  // lookup direction: the reflection vector (specular) or the normal (diffuse)
    float v[3];                    // cotangent of the sampled colour
    float dmip_dr, f;
    int l0, l1;                    // l1 < 0: one level
    bool wo_const, clamped;
};

__device__ __forceinline__ void shade_pair_pre(const float* mean, const float* normal, const float* kd, const float* ks, const float* cam_pos,
                                               float min_roughness, float max_metallic, int mode, const EnvDev& env, const float* g, PairPre& p)
{
    // the forward's own expressions (shade_one), contraction off: they select the LUT cell, the mip levels, the face and the texels
    const float rough = ks[0] * (1.0f - min_roughness) + min_roughness;
    p.metal = ks[1] * max_metallic;
    const float vv[3] = { cam_pos[0] - mean[0], cam_pos[1] - mean[1], cam_pos[2] - mean[2] };
    p.len = sqrtf(vv[0] * vv[0] + vv[1] * vv[1] + vv[2] * vv[2]);
    if (p.len < 1e-6f) { p.wo[0] = 0.0f; p.wo[1] = 0.0f; p.wo[2] = 1.0f; p.wo_const = true; }
    else { const float l = fmaxf(p.len, 1e-6f); p.wo[0] = vv[0] / l; p.wo[1] = vv[1] / l; p.wo[2] = vv[2] / l; p.wo_const = false; }
    p.d = normal[0] * p.wo[0] + normal[1] * p.wo[1] + normal[2] * p.wo[2];
    const float ndv = fmaxf(p.d, 1e-6f);
    float fg[2];
    tex2d_linear_clamp2(env.lut, env.lut_res, env.lut_res, ndv, rough, fg, p.dfg_du, p.dfg_dv);
    p.fg0 = fg[0];
    const float mip = mip_from_roughness(rough, env.min_r, env.max_r, env.L, p.dmip_dr);
    const int L = env.L;
    const float lam = fminf(fmaxf(mip, 0.0f), (float)(L - 1));
    const int lf = (int)floorf(lam);
    const bool last = lf >= L - 1;
    p.l0 = last ? L - 1 : lf;
    p.l1 = last ? -1 : lf + 1;
    p.f = last ? 0.0f : lam - (float)p.l0;
    p.clamped = last || (mip < 0.0f || mip > (float)(L - 1));
#pragma unroll
    for (int c = 0; c < 3; ++c) p.spec[c] = (1.0f - p.metal) * 0.04f + kd[c] * p.metal;
    if (mode != GS_MODE_DIFFUSE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { p.dir[k] = 2.0f * p.d * normal[k] - p.wo[k]; p.v[k] = g[k] * (p.spec[k] * fg[0] + fg[1]); }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { p.dir[k] = normal[k]; p.v[k] = g[k] * (kd[k] * (1.0f - p.metal)); }
        p.l0 = 0; p.l1 = -1; p.f = 0.0f;
    }
}

// o = sum over the levels of w * out, v_dir = sum of w * (d s / d dir), v_mip = s1 - s0 (0 when the level was clamped)
__device__ __forceinline__ void shade_pair_post(const PairPre& p, const float* normal, const float* kd, const float* g, int mode,
                                                float min_roughness, float max_metallic, const float* o, const float* v_dir, float v_mip,
                                                float* a_mean, float* a_n, float* a_kd, float* a_ks)
{
#pragma clang fp contract(fast)
    float v_diff[3] = { 0, 0, 0 }, v_rf[3] = { 0, 0, 0 }, v_refl[3] = { 0, 0, 0 }, v_n[3] = { 0, 0, 0 };
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (mode == GS_MODE_DIFFUSE) { v_diff[c] = g[c] * o[c]; v_n[c] = v_dir[c]; }
        else { v_rf[c] = g[c] * o[c]; v_refl[c] = v_dir[c]; if (mode == GS_MODE_PBR) v_diff[c] = g[c]; }
    }
    float v_A = 0.0f, v_B = 0.0f, v_metal = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v_spec = v_rf[c] * p.fg0;
        v_A += v_rf[c] * p.spec[c];
        v_B += v_rf[c];
        a_kd[c] += v_spec * p.metal + v_diff[c] * (1.0f - p.metal);
        v_metal += v_spec * (kd[c] - 0.04f) - v_diff[c] * kd[c];
    }
    const float v_ndv = v_A * p.dfg_du[0] + v_B * p.dfg_du[1];
    const float v_rough = v_A * p.dfg_dv[0] + v_B * p.dfg_dv[1] + ((p.clamped || mode == GS_MODE_DIFFUSE) ? 0.0f : v_mip) * p.dmip_dr;
    float v_d = 2.0f * (v_refl[0] * normal[0] + v_refl[1] * normal[1] + v_refl[2] * normal[2]);
    float v_wo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { v_n[j] += 2.0f * p.d * v_refl[j]; v_wo[j] = -v_refl[j]; }
    if (p.d >= 1e-6f) v_d += v_ndv;
#pragma unroll
    for (int j = 0; j < 3; ++j) { v_n[j] += v_d * p.wo[j]; v_wo[j] += v_d * normal[j]; }
    if (!p.wo_const) {
        const float dot = p.wo[0] * v_wo[0] + p.wo[1] * v_wo[1] + p.wo[2] * v_wo[2];
        const float rl = __builtin_amdgcn_rcpf(fmaxf(p.len, 1e-6f));
#pragma unroll
        for (int j = 0; j < 3; ++j) a_mean[j] -= (v_wo[j] - p.wo[j] * dot) * rl;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) a_n[j] += v_n[j];
    a_ks[0] += v_rough * (1.0f - min_roughness); a_ks[1] += v_metal * max_metallic;
}

// scatter into an LDS-resident private copy (ds_add_f32) -- flushed once per block
__device__ __forceinline__ void cube_scatter_lds(float* lds, const CubeFp& fp, const float* g, float scale)
{
    if (!fp.valid) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (fp.idx[i] < 0) continue;
        const float w = scale * fp.w[i];
        float* p = lds + (size_t)fp.idx[i] * 3;
#ifdef GS_EXPERIMENT_NO_LDS_TEXEL_ATOMICS
        if (w == 123.456f) p[0] = g[0];                      /* timing experiment only */
#else
        atomicAdd(p, g[0] * w); atomicAdd(p + 1, g[1] * w); atomicAdd(p + 2, g[2] * w);
#endif
    }
}

#define GS_XCD_COPIES 8
static size_t lds_bytes_of(int floats) { return (size_t)floats * sizeof(float); }
static size_t shade_bwd_priv_floats(const EnvDev& e, int mode, long long* level_off, long long* base_off)
{
    long long off = 0;
    for (int l = 0; l < GS_MAX_LEVELS; ++l) level_off[l] = -1;
    *base_off = -1;
    if (mode == GS_MODE_DIFFUSE) {
        if (e.base_res > 32) { *base_off = off; off += 18ll * e.base_res * e.base_res; }
    } else {
        for (int l = 0; l < e.L; ++l)
            if (e.res[l] > 32) { level_off[l] = off; off += 18ll * e.res[l] * e.res[l]; }
    }
    return (size_t)((off + 63) / 64 * 64);
}

static int env_to_dev(const GsEnv* env, EnvDev& e)
{
    if (!env || !env->lut || !env->base || env->num_levels < 1 || env->num_levels > GS_MAX_LEVELS) return -1;
    if (env->base_res > GS_CUBE_EDGE_TABLE_MAX_R) return -1;       // gs_cube.h: the edge table is checked up to this face size
    e.lut = env->lut; e.lut_res = env->lut_res; e.base = env->base; e.base_res = env->base_res;
    e.L = env->num_levels; e.min_r = env->min_roughness; e.max_r = env->max_roughness;
    for (int l = 0; l < GS_MAX_LEVELS; ++l) {
        e.levels[l] = l < e.L ? env->levels[l] : nullptr;
        e.res[l] = l < e.L ? env->res[l] : 0;
        if (l < e.L && (!e.levels[l] || e.res[l] < 1 || e.res[l] > GS_CUBE_EDGE_TABLE_MAX_R)) return -1;
    }
    return 0;
}


// dst[i] += sum over the 8 XCD-private copies (plain loads: the producer launch has ended, its L2s are written back)
static __global__ void __launch_bounds__(256)
priv_reduce_kernel(long long n, int copies, const float* __restrict__ priv, long long stride, float* __restrict__ dst)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (int x = 0; x < copies; ++x) acc += priv[x * stride + i];
        if (acc != 0.0f) dst[i] += acc;
    }
}

// LDS / launch layout of a kernel that scatters texel gradients (shade_bwd_kernel, tail_bwd_kernel): which levels get a private
// LDS copy, where the wave staging lives, how many persistent blocks.  `eg` must already hold the caller's gradient pointers.
struct ShadeBwdPlan { int block; size_t lds_bytes; int blocks; bool use_priv; size_t priv_floats; };
static inline int shade_bwd_plan(const EnvDev& e, int mode, int N, void* ws, size_t ws_bytes, EnvGradDev& eg, ShadeBwdPlan& plan,
                                 bool force_stage = false, int force_block = 0)
{
    // LDS layout: privatise every level of at most maxres^2 texels per face (and the diffuse base) within 128 KB
    const int s_block = force_block > 0 ? force_block : 512;       // (256 / 768 / 1024-thread blocks and other LDS cut-offs: measured)
    const int s_maxres = 32;
    const int lds_budget_floats = 128 * 1024 / 4;
    int used = 0;
    eg.lds_base = -1;
    for (int l = 0; l < GS_MAX_LEVELS; ++l) eg.lds_level[l] = -1;
    if (mode == GS_MODE_DIFFUSE) {
        const int cnt = 18 * e.base_res * e.base_res;
        if (e.base_res <= s_maxres && used + cnt <= lds_budget_floats) { eg.lds_base = used; used += cnt; }
    } else {
        for (int l = e.L - 1; l >= 0; --l) {
            const int cnt = 18 * e.res[l] * e.res[l];
            if (e.res[l] <= s_maxres && used + cnt <= lds_budget_floats) { eg.lds_level[l] = used; used += cnt; }
        }
    }
    eg.lds_floats = used;
    const bool staged = true;
    (void)force_stage;
    eg.stage_off = staged ? ((used + 3) & ~3) : -1;
    const int stage_floats = staged ? (eg.stage_off - used) + (s_block / 64) * 640 : 0;
    // XCD-private accumulators for the big levels (optional workspace)
    const size_t priv_floats = shade_bwd_priv_floats(e, mode, eg.priv_level, &eg.priv_base);
    const bool use_priv = ws != nullptr && priv_floats > 0;
    if (use_priv && ws_bytes < priv_floats * sizeof(float) * GS_XCD_COPIES) { gs_set_error("shade backward: workspace too small"); return GS_ENOSPC; }
    eg.priv = use_priv ? (float*)ws : nullptr;
    eg.priv_stride = (long long)priv_floats;
    plan.block = s_block;
    plan.lds_bytes = (size_t)(used + stage_floats) * sizeof(float);
    int blocks = gs_cdiv(N > 0 ? N : 1, s_block);
    // persistent blocks when LDS copies have to be flushed at the end (one flush per block): as many as are resident at once
    const int per_cu = used > 0 ? (int)fmin(8.0, fmax(1.0, floor(160.0 * 1024.0 / (double)(plan.lds_bytes + 1024)))) : 8;
    const int max_blocks = used > 0 ? 256 * per_cu : 2048;
    plan.blocks = blocks > max_blocks ? max_blocks : blocks;
    plan.use_priv = use_priv;
    plan.priv_floats = priv_floats;
    return GS_OK;
}
