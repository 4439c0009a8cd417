// gs_shade.hip -- S1..S4: per-Gaussian split-sum PBR shading (forward + recompute-backward) and tone mapping.
//   S1 RenderableAttrs.splat arithmetic         rfstudio/model/geosplat.py:80-122
//   S2 FG-LUT bilinear/clamp lookup             rfstudio/model/geosplat.py:93-98 (nvdiffrast dr.texture 2-D)
//   S3 TextureSplitSum.sample                   rfstudio/graphics/_mesh/_texture.py:571-613
//      (cube 'linear' on base; cube 'linear-mipmap-linear' on the specular pyramid, level = f(roughness))
//   S4 _tone_mapping_naive/_aces                rfstudio/model/geosplat.py:474-480
//
// The reference runs ~25 elementwise launches plus three texture launches per view and lets autograd
// keep ~20 N-sized temporaries; here it is ONE streaming kernel per direction: 44 B/Gaussian in
// (means, normals, kd, ks), 12 B out, texture taps served by L2 / Infinity Cache (the whole pyramid is
// ~25 MB).  The backward recomputes the forward (no saved state) and scatters texel gradients with fp32
// atomics.  HBM-bound by design; cube-map texel semantics are documented in oracle/gs_oracle_shade.c.
#include "gs_common.h"
#include <stdlib.h>
#include <math.h>

// Contraction OFF for the whole file: texel / LUT-cell / lobe-membership selection are discontinuous in the
// coordinates, so the coordinates are computed in the same one-rounding-per-operation order as the CPU oracle
// (these kernels are memory-bound; the lost FMAs cost nothing measurable).
#pragma clang fp contract(off)
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

__global__ void __launch_bounds__(256)
shade_fwd_kernel(int N, const float* __restrict__ means, const float* __restrict__ normals,
                 const float* __restrict__ kd, const float* __restrict__ ks, const float* __restrict__ cam_pos,
                 float min_roughness, float max_metallic, int mode, EnvDev env, float* __restrict__ colors)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float mean[3] = { means[3 * (size_t)n], means[3 * (size_t)n + 1], means[3 * (size_t)n + 2] };
    const float nrm[3] = { normals[3 * (size_t)n], normals[3 * (size_t)n + 1], normals[3 * (size_t)n + 2] };
    const float kdn[3] = { kd[3 * (size_t)n], kd[3 * (size_t)n + 1], kd[3 * (size_t)n + 2] };
    const float2 ks2 = *reinterpret_cast<const float2*>(ks + 2 * (size_t)n);
    const float ksn[2] = { ks2.x, ks2.y };
    const float cp[3] = { cam_pos[0], cam_pos[1], cam_pos[2] };
    ShadeTmp t;
    float color[3];
    shade_one<false>(mean, nrm, kdn, ksn, cp, min_roughness, max_metallic, mode, env, color, t);
    colors[3 * (size_t)n] = color[0]; colors[3 * (size_t)n + 1] = color[1]; colors[3 * (size_t)n + 2] = color[2];
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

// Persistent blocks (one per CU: the private texel-gradient copies of the <=32^2 levels take ~90 KB of the
// 160 KB LDS).  2 M Gaussians send ~13 M atomics at the 4 608 floats of the 16^2 level alone: in HBM/L2 that
// serialises per address (5.5 ms per view measured); in LDS it is a ds_add_f32 and the block flushes its
// copy once at the end.
// Block size and the largest LDS-privatised level are run-time choices (GEOSPLAT_SHADE_BWD_BLOCK / GEOSPLAT_SHADE_LDS_MAXRES):
// the round-1 shape -- 1024-thread blocks, one per CU, levels <= 32^2 in 92 KB of LDS -- caps the kernel at 128 VGPRs and it
// SPILLS (264 bytes of scratch per lane, -Rpass-analysis): the recomputed forward then runs at a fifth of the forward
// kernel's rate.  Smaller blocks lift the cap (256 registers at 2 waves per SIMD).  Measured at the bench workload (scripts/shade_ab.py):
// 1024 threads / <= 32^2 in LDS 470 us (round 1); 512 / 32^2 389 us (default); 256 / 32^2 527 (4 waves per CU); 512 or 256 with only
// the 16^2 level in LDS 930-960 (the 32^2 level through memory-side atomics: 2 800 requests per cache line); no LDS copies 1 570.
// Removal experiment at 512 / 32^2: 349 us without the LDS atomics, 333 without the global ones, 254 without either.
template <bool PRIV, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
shade_bwd_kernel(int N, const float* __restrict__ means, const float* __restrict__ normals,
                 const float* __restrict__ kd, const float* __restrict__ ks, const float* __restrict__ cam_pos,
                 float min_roughness, float max_metallic, int mode, EnvDev env, const float* __restrict__ v_colors,
                 float* __restrict__ v_means, float* __restrict__ v_normals, float* __restrict__ v_kd,
                 float* __restrict__ v_ks, EnvGradDev eg, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) float s_grad[];
    for (int i = threadIdx.x; i < eg.lds_floats; i += blockDim.x) s_grad[i] = 0.0f;
    __syncthreads();
    // this block's XCD-private accumulator copy (device-scope fp32 atomics are resolved at the memory side of the
    // fabric -- a 32-byte write-through each, 0.6 GB per view here; XCD-local ones stay in the 4 MiB L2)
    float* const priv = PRIV ? eg.priv + (long long)gs_xcc_id() * eg.priv_stride : nullptr;
    // wave-private commit staging behind the private texel copies (640 floats per wave, see wave_commit6_lds); -1 = permute form
    float* const stage = eg.stage_off >= 0 ? s_grad + eg.stage_off + (threadIdx.x >> 6) * 640 : nullptr;
    const float cp[3] = { cam_pos[0], cam_pos[1], cam_pos[2] };
    // wave-uniform trip count: every lane of a wave reaches the wave-aggregated scatter together
    const int n_iter = (N + (int)(gridDim.x * blockDim.x) - 1) / (int)(gridDim.x * blockDim.x);
    for (int it = 0; it < n_iter; ++it) {
        const int n = it * (int)(gridDim.x * blockDim.x) + blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = n < N;
        bool scatter = false;              // this lane has texel gradients to commit
        ShadeTmp t;
        float v_ls[3] = { 0, 0, 0 }, v_ld[3] = { 0, 0, 0 };
        if (live) {
        const float mean[3] = { means[3 * (size_t)n], means[3 * (size_t)n + 1], means[3 * (size_t)n + 2] };
        const float normal[3] = { normals[3 * (size_t)n], normals[3 * (size_t)n + 1], normals[3 * (size_t)n + 2] };
        const float kdn[3] = { kd[3 * (size_t)n], kd[3 * (size_t)n + 1], kd[3 * (size_t)n + 2] };
        const float2 ks2 = *reinterpret_cast<const float2*>(ks + 2 * (size_t)n);
        const float ksn[2] = { ks2.x, ks2.y };
        const float g[3] = { v_colors[3 * (size_t)n], v_colors[3 * (size_t)n + 1], v_colors[3 * (size_t)n + 2] };
        if (g[0] == 0.0f && g[1] == 0.0f && g[2] == 0.0f) {
            // Gaussian that reached no pixel (culled, hidden behind the opaque front layer, ...): every gradient
            // of the shading is exactly zero -- skip the texture taps and the texel atomics
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (!accumulate) { v_means[3 * (size_t)n + k] = 0.0f; v_normals[3 * (size_t)n + k] = 0.0f; v_kd[3 * (size_t)n + k] = 0.0f; }
            }
            if (!accumulate) *reinterpret_cast<float2*>(v_ks + 2 * (size_t)n) = make_float2(0.0f, 0.0f);
        } else {
        float color[3];
        shade_one<true>(mean, normal, kdn, ksn, cp, min_roughness, max_metallic, mode, env, color, t);
        scatter = true;

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
        float o_mean[3] = { 0, 0, 0 };
        if (!t.wo_const) {
            const float dot = t.wo[0] * v_wo[0] + t.wo[1] * v_wo[1] + t.wo[2] * v_wo[2];
            const float l = fmaxf(t.len, 1e-6f);
#pragma unroll
            for (int k = 0; k < 3; ++k) o_mean[k] = -((v_wo[k] - t.wo[k] * dot) / l);
        }
        if (accumulate) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v_means[3 * (size_t)n + k] += o_mean[k];
                v_normals[3 * (size_t)n + k] += v_n[k];
                v_kd[3 * (size_t)n + k] += o_kd[k];
            }
            float2 o = *reinterpret_cast<float2*>(v_ks + 2 * (size_t)n);
            o.x += v_rough * (1.0f - min_roughness); o.y += v_metal * max_metallic;
            *reinterpret_cast<float2*>(v_ks + 2 * (size_t)n) = o;
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                v_means[3 * (size_t)n + k] = o_mean[k];
                v_normals[3 * (size_t)n + k] = v_n[k];
                v_kd[3 * (size_t)n + k] = o_kd[k];
            }
            *reinterpret_cast<float2*>(v_ks + 2 * (size_t)n) = make_float2(v_rough * (1.0f - min_roughness), v_metal * max_metallic);
        }

        }   // non-zero upstream gradient
        }   // live
        // ---- texel gradients: LDS-private copies for the small levels, wave-aggregated atomics for the rest
        if (mode != GS_MODE_DIFFUSE) {
            const int l0 = scatter ? t.ls.l0 : 0, l1 = scatter ? t.ls.l1 : -1;
            const float w0 = (l1 < 0) ? 1.0f : 1.0f - t.ls.f;
            const bool lds0 = scatter && eg.lds_level[l0] >= 0;
            if (lds0) cube_scatter_lds(s_grad + eg.lds_level[l0], t.ls.fp0, v_ls, w0);
            cube_scatter_wave<PRIV>(scatter && !lds0 ? (PRIV ? priv + eg.priv_level[l0] : eg.levels[l0]) : nullptr, t.ls.fp0,
                                    v_ls, w0, scatter && !lds0, stage);
            const bool has1 = scatter && l1 >= 0;
            const bool lds1 = has1 && eg.lds_level[has1 ? l1 : 0] >= 0;
            if (lds1) cube_scatter_lds(s_grad + eg.lds_level[l1], t.ls.fp1, v_ls, t.ls.f);
            cube_scatter_wave<PRIV>(has1 && !lds1 ? (PRIV ? priv + eg.priv_level[has1 ? l1 : 0] : eg.levels[l1]) : nullptr,
                                    t.ls.fp1, v_ls, t.ls.f, has1 && !lds1, stage);
        } else {
            const bool ldsb = scatter && eg.lds_base >= 0;
            if (ldsb) cube_scatter_lds(s_grad + eg.lds_base, t.ld_fp, v_ld, 1.0f);
            cube_scatter_wave<PRIV>(scatter && !ldsb ? (PRIV ? priv + eg.priv_base : eg.base) : nullptr, t.ld_fp, v_ld, 1.0f,
                                    scatter && !ldsb, stage);
        }
    }
    // ---- flush the private copies
    __syncthreads();
    if (eg.lds_base >= 0) {
        const int cnt = 18 * env.base_res * env.base_res;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const float v = s_grad[eg.lds_base + i];
            if (v != 0.0f) gs_atomic_add(eg.base + i, v);
        }
    }
    for (int l = 0; l < env.L; ++l) {
        if (eg.lds_level[l] < 0) continue;
        const int cnt = 18 * env.res[l] * env.res[l];
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const float v = s_grad[eg.lds_level[l] + i];
            if (v != 0.0f) gs_atomic_add(eg.levels[l] + i, v);
        }
    }
}

// dst[i] += sum over the 8 XCD-private copies (plain loads: the producer launch has ended, its L2s are written back)
__global__ void __launch_bounds__(256)
priv_reduce_kernel(long long n, int copies, const float* __restrict__ priv, long long stride, float* __restrict__ dst)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        for (int x = 0; x < copies; ++x) acc += priv[x * stride + i];
        if (acc != 0.0f) dst[i] += acc;
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
    e.lut = env->lut; e.lut_res = env->lut_res; e.base = env->base; e.base_res = env->base_res;
    e.L = env->num_levels; e.min_r = env->min_roughness; e.max_r = env->max_roughness;
    for (int l = 0; l < GS_MAX_LEVELS; ++l) {
        e.levels[l] = l < e.L ? env->levels[l] : nullptr;
        e.res[l] = l < e.L ? env->res[l] : 0;
        if (l < e.L && (!e.levels[l] || e.res[l] < 1)) return -1;
    }
    return 0;
}

extern "C" int gs_shade_fwd(int N, const float* means, const float* normals, const float* kd, const float* ks,
                            const float* cam_pos, float min_roughness, float max_metallic, int mode,
                            const GsEnv* env, float* colors, void* stream)
{
    GS_CHECK_ARG(N >= 0 && mode >= 0 && mode <= 2, "bad N or mode");
    EnvDev e;
    GS_CHECK_ARG(env_to_dev(env, e) == 0, "bad GsEnv");
    if (N == 0) return GS_OK;
    hipLaunchKernelGGL(shade_fwd_kernel, dim3(gs_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, N, means, normals,
                       kd, ks, cam_pos, min_roughness, max_metallic, mode, e, colors);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" size_t gs_shade_bwd_ws_bytes(const GsEnv* env, int mode)
{
    EnvDev e;
    if (env_to_dev(env, e) != 0) return 0;
    long long lo[GS_MAX_LEVELS], bo;
    return shade_bwd_priv_floats(e, mode, lo, &bo) * sizeof(float) * GS_XCD_COPIES;
}

extern "C" int gs_shade_bwd(int N, const float* means, const float* normals, const float* kd, const float* ks,
                            const float* cam_pos, float min_roughness, float max_metallic, int mode,
                            const GsEnv* env, const float* v_colors, float* v_means, float* v_normals,
                            float* v_kd, float* v_ks, const GsEnvGrad* env_grad, int accumulate, void* ws,
                            size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(N >= 0 && mode >= 0 && mode <= 2, "bad N or mode");
    EnvDev e;
    GS_CHECK_ARG(env_to_dev(env, e) == 0, "bad GsEnv");
    GS_CHECK_ARG(env_grad != nullptr, "env_grad must not be NULL");
    EnvGradDev eg;
    eg.base = env_grad->base;
    for (int l = 0; l < GS_MAX_LEVELS; ++l) eg.levels[l] = l < e.L ? env_grad->levels[l] : nullptr;
    if (mode == GS_MODE_DIFFUSE) GS_CHECK_ARG(eg.base != nullptr, "env_grad->base required in diffuse mode");
    else for (int l = 0; l < e.L; ++l) GS_CHECK_ARG(eg.levels[l] != nullptr, "env_grad->levels[l] required");
    // LDS layout: privatise every level of at most maxres^2 texels per face (and the diffuse base) within 128 KB
    static const int s_block = [] { const char* v = getenv("GEOSPLAT_SHADE_BWD_BLOCK"); const int b = v ? atoi(v) : 512; return (b == 256 || b == 768 || b == 1024) ? b : 512; }();
    static const int s_maxres = [] { const char* v = getenv("GEOSPLAT_SHADE_LDS_MAXRES"); const int r = v ? atoi(v) : 32; return r; }();
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
    static const bool s_stage = [] { const char* v = getenv("GEOSPLAT_SHADE_COMMIT_LDS"); return !(v && v[0] == '0'); }();
    eg.stage_off = s_stage ? ((used + 3) & ~3) : -1;
    const int stage_floats = s_stage ? (eg.stage_off - used) + (s_block / 64) * 640 : 0;
    // XCD-private accumulators for the big levels (optional workspace)
    const size_t priv_floats = shade_bwd_priv_floats(e, mode, eg.priv_level, &eg.priv_base);
    const bool use_priv = ws != nullptr && priv_floats > 0;
    if (use_priv && ws_bytes < priv_floats * sizeof(float) * GS_XCD_COPIES) { gs_set_error("gs_shade_bwd: workspace too small"); return GS_ENOSPC; }
    eg.priv = use_priv ? (float*)ws : nullptr;
    eg.priv_stride = (long long)priv_floats;
    if (N == 0) return GS_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds_bytes = (size_t)(used + stage_floats) * sizeof(float);
    int blocks = gs_cdiv(N, s_block);
    // persistent blocks when LDS copies have to be flushed at the end (one flush per block): as many as are resident at once
    const int per_cu = used > 0 ? (int)fmin(8.0, fmax(1.0, floor(160.0 * 1024.0 / (double)(lds_bytes + 1024)))) : 8;
    const int max_blocks = used > 0 ? 256 * per_cu : 2048;
    if (blocks > max_blocks) blocks = max_blocks;
    if (use_priv) {
        GS_CHECK_HIP(gs_zero_async(ws, priv_floats * sizeof(float) * GS_XCD_COPIES, s));
#define GS_SHADE_LAUNCH(P, B)                                                                                                   \
        do {                                                                                                                    \
            GS_CHECK_HIP(hipFuncSetAttribute((const void*)shade_bwd_kernel<P, B>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
            hipLaunchKernelGGL((shade_bwd_kernel<P, B>), dim3(blocks), dim3(B), lds_bytes, s, N, means, normals, kd, ks, cam_pos,     \
                               min_roughness, max_metallic, mode, e, v_colors, v_means, v_normals, v_kd, v_ks, eg, accumulate);   \
            GS_CHECK_LAUNCH();                                                                                                  \
        } while (0)
        if (s_block == 1024) GS_SHADE_LAUNCH(true, 1024); else if (s_block == 768) GS_SHADE_LAUNCH(true, 768); else if (s_block == 512) GS_SHADE_LAUNCH(true, 512); else GS_SHADE_LAUNCH(true, 256);
        // fold the 8 copies into the caller's gradient buffers, level by level
        for (int l = -1; l < e.L; ++l) {
            const long long off = l < 0 ? eg.priv_base : eg.priv_level[l];
            if (off < 0) continue;
            const int R = l < 0 ? e.base_res : e.res[l];
            float* dst = l < 0 ? eg.base : eg.levels[l];
            const long long n = 18ll * R * R;
            hipLaunchKernelGGL(priv_reduce_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s, n,
                               GS_XCD_COPIES, (const float*)ws + off, (long long)priv_floats, dst);
            GS_CHECK_LAUNCH();
        }
    } else {
        if (s_block == 1024) GS_SHADE_LAUNCH(false, 1024); else if (s_block == 768) GS_SHADE_LAUNCH(false, 768); else if (s_block == 512) GS_SHADE_LAUNCH(false, 512); else GS_SHADE_LAUNCH(false, 256);
#undef GS_SHADE_LAUNCH
    }
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------
// S4 tone mapping
#include "gs_tone.h"

__global__ void __launch_bounds__(256)
tonemap_fwd_kernel(int64_t P, int mode, const float4* __restrict__ rgba, const float* __restrict__ exposure,
                   float4* __restrict__ out)
{
    const float e = exposure[0];
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = rgba[p];
        // 'none' is `render_rgba * exposure` (rfstudio/model/geosplat.py:123-124): all four channels, alpha included
        out[p] = make_float4(tone_fwd(mode, v.x * e), tone_fwd(mode, v.y * e), tone_fwd(mode, v.z * e),
                             mode == GS_TONE_NONE ? v.w * e : v.w);
    }
}

__global__ void __launch_bounds__(256)
tonemap_bwd_kernel(int64_t P, int mode, const float4* __restrict__ rgba, const float* __restrict__ exposure,
                   const float4* __restrict__ v_out, float4* __restrict__ v_rgba, float* __restrict__ v_exposure)
{
    const float e = exposure[0];
    float ve = 0.0f;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = rgba[p], g = v_out[p];
        const float gx = g.x * tone_grad(mode, v.x * e), gy = g.y * tone_grad(mode, v.y * e), gz = g.z * tone_grad(mode, v.z * e);
        v_rgba[p] = make_float4(gx * e, gy * e, gz * e, mode == GS_TONE_NONE ? g.w * e : g.w);
        ve += gx * v.x + gy * v.y + gz * v.z + (mode == GS_TONE_NONE ? g.w * v.w : 0.0f);
    }
    ve = gs_wave_sum(ve);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = ve;
    __syncthreads();
    if (threadIdx.x == 0) gs_atomic_add(v_exposure, s[0] + s[1] + s[2] + s[3]);
}

// The same pair on the rasterizer's own layout: render [P,3] + alphas [P] in, image [P,4] out; and v_render [P,3] / v_alphas [P]
// out of the backward -- no rgba concatenation, no strided copies in between (the engine's per-view glue).
__global__ void __launch_bounds__(256)
tonemap_fwd3_kernel(int64_t P, int mode, const float* __restrict__ render, const float* __restrict__ alphas,
                    const float* __restrict__ exposure, float4* __restrict__ out)
{
    const float e = exposure[0];
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const float r = render[3 * p], g = render[3 * p + 1], b = render[3 * p + 2], a = alphas[p];
        out[p] = make_float4(tone_fwd(mode, r * e), tone_fwd(mode, g * e), tone_fwd(mode, b * e), mode == GS_TONE_NONE ? a * e : a);
    }
}

__global__ void __launch_bounds__(256)
tonemap_bwd3_kernel(int64_t P, int mode, const float* __restrict__ render, const float* __restrict__ alphas,
                    const float* __restrict__ exposure, const float4* __restrict__ v_out, float* __restrict__ v_render,
                    float* __restrict__ v_alphas, float* __restrict__ v_exposure)
{
    const float e = exposure[0];
    float ve = 0.0f;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const float r = render[3 * p], gch = render[3 * p + 1], b = render[3 * p + 2], a = alphas[p];
        const float4 g = v_out[p];
        const float gx = g.x * tone_grad(mode, r * e), gy = g.y * tone_grad(mode, gch * e), gz = g.z * tone_grad(mode, b * e);
        v_render[3 * p] = gx * e; v_render[3 * p + 1] = gy * e; v_render[3 * p + 2] = gz * e;
        v_alphas[p] = mode == GS_TONE_NONE ? g.w * e : g.w;
        ve += gx * r + gy * gch + gz * b + (mode == GS_TONE_NONE ? g.w * a : 0.0f);
    }
    ve = gs_wave_sum(ve);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = ve;
    __syncthreads();
    if (threadIdx.x == 0) gs_atomic_add(v_exposure, s[0] + s[1] + s[2] + s[3]);
}

extern "C" int gs_tonemap_fwd3(int64_t P, int mode, const float* render, const float* alphas, const float* exposure, float* out,
                               void* stream)
{
    GS_CHECK_ARG(P >= 0 && mode >= 0 && mode <= 2, "bad P or mode");
    if (P == 0) return GS_OK;
    const int blocks = (int)((P + 255) / 256 < 2048 ? (P + 255) / 256 : 2048);
    hipLaunchKernelGGL(tonemap_fwd3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, mode, render, alphas, exposure, (float4*)out);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_tonemap_bwd3(int64_t P, int mode, const float* render, const float* alphas, const float* exposure,
                               const float* v_out, float* v_render, float* v_alphas, float* v_exposure, int accumulate, void* stream)
{
    GS_CHECK_ARG(P >= 0 && mode >= 0 && mode <= 2, "bad P or mode");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) GS_CHECK_HIP(gs_zero_async(v_exposure, sizeof(float), s));
    if (P == 0) return GS_OK;
    const int blocks = (int)((P + 255) / 256 < 1024 ? (P + 255) / 256 : 1024);
    hipLaunchKernelGGL(tonemap_bwd3_kernel, dim3(blocks), dim3(256), 0, s, P, mode, render, alphas, exposure, (const float4*)v_out,
                       v_render, v_alphas, v_exposure);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_tonemap_fwd(int64_t P, int mode, const float* rgba, const float* exposure, float* out, void* stream)
{
    GS_CHECK_ARG(P >= 0 && mode >= 0 && mode <= 2, "bad P or mode");
    if (P == 0) return GS_OK;
    const int blocks = (int)((P + 255) / 256 < 2048 ? (P + 255) / 256 : 2048);
    hipLaunchKernelGGL(tonemap_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, mode,
                       (const float4*)rgba, exposure, (float4*)out);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_tonemap_bwd(int64_t P, int mode, const float* rgba, const float* exposure, const float* v_out,
                              float* v_rgba, float* v_exposure, int accumulate, void* stream)
{
    GS_CHECK_ARG(P >= 0 && mode >= 0 && mode <= 2, "bad P or mode");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) GS_CHECK_HIP(gs_zero_async(v_exposure, sizeof(float), s));
    if (P == 0) return GS_OK;
    const int blocks = (int)((P + 255) / 256 < 1024 ? (P + 255) / 256 : 1024);
    hipLaunchKernelGGL(tonemap_bwd_kernel, dim3(blocks), dim3(256), 0, s, P, mode, (const float4*)rgba, exposure,
                       (const float4*)v_out, (float4*)v_rgba, v_exposure);
    GS_CHECK_LAUNCH();
    return GS_OK;
}
