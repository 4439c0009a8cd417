// gs_project_dev.h -- device-side pieces of the projection stage (A1 / A7) shared by gs_project.hip (the rasterization() call
// shape) and gs_front.hip (the engine's fused front / tail kernels): camera load, the canonical-order forward projection
// (bit-exact twin of the oracle's project_one), tile ranges, the chained-scan descriptor helpers and the projection backward
// of ONE Gaussian.  Semantics: gsplat 1.4 fully_fused_projection(packed) as reached from rfstudio/model/gsplat.py:334-355.
#pragma once
#include "gs_common.h"

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

__device__ __forceinline__ long long wave_sum_i64(long long v)          // (counts: non-negative, far below 2^62)
{
    return (long long)gs_wave_sum_u64((unsigned long long)v);
}

// ---------------------------------------------------------------------------------------------------
// A7 for ONE Gaussian: chain {v_means2d, v_conics, v_opacity(packed), v_depth} back to {v_mean, v_quat, v_scale, v_opacity};
// the forward intermediates are recomputed (no saved state).  Shared by project_bwd_kernel (gs_project.hip) and the engine's
// fused tail kernel (gs_front.hip).  (ia, ib, ic) = the conic, comp = the anti-alias compensation of the forward.
struct ProjGrad { float mean[3], quat[4], scale[3], op; };

__device__ __forceinline__ void project_bwd_one(const GsCam& cam, const float* mean, float4 q4, const float* scale, float opacity,
                                                float Wf, float Hf, float eps2d, float ia, float ib, float ic, float comp,
                                                float vm2x, float vm2y, float v_ca, float v_cb, float v_cc, float v_op, float v_depth,
                                                ProjGrad& g)
{
    // a gradient, compared to 1e-4 and not bit for bit: contraction ON (the file default is off), reciprocals by v_rcp_f32
#pragma clang fp contract(fast)
    g.op = v_op * comp;
    const float v_comp = v_op * opacity;

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
    const float ga = v_ca, gb = 0.5f * v_cb, gc = v_cc;
    const float p00 = ia * ga + ib * gb, p01 = ia * gb + ib * gc;
    const float p10 = ib * ga + ic * gb, p11 = ib * gb + ic * gc;
    float G[4] = { -(p00 * ia + p01 * ib), -(p00 * ib + p01 * ic), -(p10 * ia + p11 * ib), -(p10 * ib + p11 * ic) };
    {   // compensation vjp
        const float det_conic = ia * ic - ib * ib;
        const float v_sqr_comp = v_comp * 0.5f * __builtin_amdgcn_rcpf(comp + 1e-6f);
        const float om = 1.0f - comp * comp;
        G[0] += v_sqr_comp * (om * ia - eps2d * det_conic);
        G[1] += v_sqr_comp * (om * ib);
        G[2] += v_sqr_comp * (om * ib);
        G[3] += v_sqr_comp * (om * ic - eps2d * det_conic);
    }
    // perspective projection vjp
    const float x = mc[0], y = mc[1], z = mc[2];
    const float rfx = __builtin_amdgcn_rcpf(cam.fx), rfy = __builtin_amdgcn_rcpf(cam.fy);
    const float tan_fovx = 0.5f * Wf * rfx, tan_fovy = 0.5f * Hf * rfy;
    const float lim_x_pos = (Wf - cam.cx) * rfx + 0.3f * tan_fovx;
    const float lim_x_neg = cam.cx * rfx + 0.3f * tan_fovx;
    const float lim_y_pos = (Hf - cam.cy) * rfy + 0.3f * tan_fovy;
    const float lim_y_neg = cam.cy * rfy + 0.3f * tan_fovy;
    const float rz = __builtin_amdgcn_rcpf(z), rz2 = rz * rz, rz3 = rz2 * rz;
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
    v_mc[2] += v_depth;

    const float* R = cam.R;
#pragma unroll
    for (int i = 0; i < 3; ++i) g.mean[i] = R[0 * 3 + i] * v_mc[0] + R[1 * 3 + i] * v_mc[1] + R[2 * 3 + i] * v_mc[2];
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
    for (int j = 0; j < 3; ++j) g.scale[j] = Rq[0 * 3 + j] * v_M[0 * 3 + j] + Rq[1 * 3 + j] * v_M[1 * 3 + j] + Rq[2 * 3 + j] * v_M[2 * 3 + j];

    const float inv = __builtin_amdgcn_rsqf(q4.y * q4.y + q4.z * q4.z + q4.w * q4.w + q4.x * q4.x);
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
    for (int k = 0; k < 4; ++k) g.quat[k] = (vqn[k] - dotp * qn[k]) * inv;
}
