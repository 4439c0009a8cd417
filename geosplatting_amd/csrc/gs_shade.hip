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
#include "gs_shade_dev.h"

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

// Persistent blocks (one per CU: the private texel-gradient copies of the <=32^2 levels take ~90 KB of the
// 160 KB LDS).  2 M Gaussians send ~13 M atomics at the 4 608 floats of the 16^2 level alone: in HBM/L2 that
// serialises per address (5.5 ms per view measured); in LDS it is a ds_add_f32 and the block flushes its
// copy once at the end.
// Block size (512) and the largest LDS-privatised level (32^2) were run-time choices while they were being measured:
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
    ShadeBwdPlan plan;
    { const int rc = shade_bwd_plan(e, mode, N, ws, ws_bytes, eg, plan); if (rc != GS_OK) return rc; }
    if (N == 0) return GS_OK;
    hipStream_t s = (hipStream_t)stream;
    const int s_block = plan.block, blocks = plan.blocks;
    const size_t lds_bytes = plan.lds_bytes, priv_floats = plan.priv_floats;
    const bool use_priv = plan.use_priv;
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

// self-test hook of the cube edge table (gs_cube.h resolve_texel): every (face, edge, t) of an R x R face against the float rule
__global__ void __launch_bounds__(256) selftest_cube_edges_kernel(int R, unsigned long long* mismatches)
{
    const long long total = 24ll * R;
    unsigned long long bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % R), e = (int)((i / R) & 3), s = (int)(i / (4ll * R));
        const int ix = e == 0 ? -1 : (e == 1 ? R : t), iy = e == 2 ? -1 : (e == 3 ? R : t);
        bad += resolve_texel(s, ix, iy, R) != resolve_texel_reproject(s, ix, iy, R);
    }
    if (bad) atomicAdd(mismatches, bad);
}

extern "C" int gs_selftest_cube_edges(int R, uint64_t* mismatches_dev, void* stream)
{
    GS_CHECK_ARG(mismatches_dev != nullptr && R >= 1 && R <= GS_CUBE_EDGE_TABLE_MAX_R, "bad size");
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_HIP(gs_zero_async(mismatches_dev, sizeof(uint64_t), s));
    hipLaunchKernelGGL(selftest_cube_edges_kernel, dim3(gs_cdiv(24 * R, 256)), dim3(256), 0, s, R, (unsigned long long*)mismatches_dev);
    GS_CHECK_LAUNCH();
    return GS_OK;
}
