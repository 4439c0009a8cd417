// gs_mesh.hip -- SURVEY section 8f rank 1: MGAdapter, the step immediately before the render path
// (rfstudio/model/geosplat.py:378-472): every mesh face becomes 6 flat Gaussians (two rings of three),
// means / log-scales / wxyz quaternions / interpolated shading normals.
//
// One thread per face; outputs are laid out like the reference's Splats.cat of the six parts:
// row = part * F + face.  The backward does not hand-derive the chain (cross products, two normalisations,
// rot2quat with its value-dependent branch): the SAME templated forward is evaluated on dual numbers
// (value + one tangent), once per input coordinate (18 per face), and the dot product of the output tangents with
// the incoming gradients is that coordinate's gradient -- exact forward-mode AD, ~18x the (tiny) forward cost.
// Vertex gradients are accumulated with fp32 atomics (a vertex is shared by ~6 faces).
#include "gs_common.h"

struct Dual {
    float v, d;
};
__device__ __forceinline__ Dual mk(float v, float d = 0.0f) { Dual r; r.v = v; r.d = d; return r; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return mk(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return mk(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return mk(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return mk(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { const float q = a.v / b.v; return mk(q, (a.d - q * b.d) / b.v); }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return mk(a * b.v, a * b.d); }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return mk(a.v * b, a.d * b); }
__device__ __forceinline__ Dual operator/(Dual a, float b) { return mk(a.v / b, a.d / b); }
__device__ __forceinline__ Dual operator+(float a, Dual b) { return mk(a + b.v, b.d); }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return mk(a - b.v, -b.d); }

template <typename T> struct Num;
template <> struct Num<float> {
    static __device__ __forceinline__ float val(float a) { return a; }
    static __device__ __forceinline__ float cst(float a) { return a; }
    static __device__ __forceinline__ float sqrt_(float a) { return sqrtf(a); }
    static __device__ __forceinline__ float log_(float a) { return logf(a); }
    static __device__ __forceinline__ float clamp_min(float a, float c) { return a < c ? c : a; }
};
template <> struct Num<Dual> {
    static __device__ __forceinline__ float val(Dual a) { return a.v; }
    static __device__ __forceinline__ Dual cst(float a) { return mk(a, 0.0f); }
    static __device__ __forceinline__ Dual sqrt_(Dual a) { const float s = sqrtf(a.v); return mk(s, a.d * 0.5f / s); }
    static __device__ __forceinline__ Dual log_(Dual a) { return mk(logf(a.v), a.d / a.v); }
    static __device__ __forceinline__ Dual clamp_min(Dual a, float c) { return a.v < c ? mk(c, 0.0f) : a; }
};

template <typename T> __device__ __forceinline__ void cross3(const T* a, const T* b, T* o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T> __device__ __forceinline__ T norm3(const T* a) { return Num<T>::sqrt_(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// safe_normalize (rfstudio/graphics/math.py:119-125): |v| < 1e-6 -> constant (0,0,1)
template <typename T> __device__ __forceinline__ void safe_normalize3(const T* v, T* o)
{
    const T l = norm3(v);
    if (Num<T>::val(l) < 1e-6f) { o[0] = Num<T>::cst(0.0f); o[1] = Num<T>::cst(0.0f); o[2] = Num<T>::cst(1.0f); return; }
    const T lc = Num<T>::clamp_min(l, 1e-6f);
    o[0] = v[0] / lc; o[1] = v[1] / lc; o[2] = v[2] / lc;
}

// rot2quat (rfstudio/graphics/math.py:246-278): best-conditioned of the four candidates, wxyz
template <typename T> __device__ void rot2quat(const T m[3][3], T* q)
{
    const T one = Num<T>::cst(1.0f);
    T qq[4] = { one + m[0][0] + m[1][1] + m[2][2], one + m[0][0] - m[1][1] - m[2][2],
                one - m[0][0] + m[1][1] - m[2][2], one - m[0][0] - m[1][1] + m[2][2] };
    T qa[4];
    int best = 0; float bestv = -1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        qa[i] = Num<T>::val(qq[i]) > 0.0f ? Num<T>::sqrt_(qq[i]) : Num<T>::cst(0.0f);
        if (Num<T>::val(qa[i]) > bestv) { bestv = Num<T>::val(qa[i]); best = i; }
    }
    T c[4];
    if (best == 0)      { c[0] = qa[0] * qa[0];        c[1] = m[2][1] - m[1][2]; c[2] = m[0][2] - m[2][0]; c[3] = m[1][0] - m[0][1]; }
    else if (best == 1) { c[0] = m[2][1] - m[1][2];    c[1] = qa[1] * qa[1];     c[2] = m[1][0] + m[0][1]; c[3] = m[0][2] + m[2][0]; }
    else if (best == 2) { c[0] = m[0][2] - m[2][0];    c[1] = m[1][0] + m[0][1]; c[2] = qa[2] * qa[2];     c[3] = m[1][2] + m[2][1]; }
    else                { c[0] = m[1][0] - m[0][1];    c[1] = m[2][0] + m[0][2]; c[2] = m[2][1] + m[1][2]; c[3] = qa[3] * qa[3]; }
    const T den = 2.0f * Num<T>::clamp_min(qa[best], 0.1f);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = c[i] / den;
}

// bary2gs (rfstudio/model/geosplat.py:390-424)
template <typename T>
__device__ void bary2gs(const T* a0, const T* a1, T area, const T* fnn, float s_ratio, T* mean, T* scale, T* quat)
{
    const float g_scale_ratio = 1.6f;
    T mr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { mean[k] = (a0[k] + a1[k]) / 2.0f; mr[k] = a1[k] - mean[k]; }
    const T ms = Num<T>::clamp_min(norm3(mr), 1e-10f);
    const T mins = area / 4.0f / ms;
#pragma unroll
    for (int k = 0; k < 3; ++k) mr[k] = mr[k] / ms;
    scale[0] = Num<T>::log_((g_scale_ratio * s_ratio) * ms);
    scale[1] = Num<T>::log_((g_scale_ratio / s_ratio) * mins);
    scale[2] = Num<T>::cst(-10.0f);
    T minr[3];
    cross3(fnn, mr, minr);
    T R[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { R[k][0] = mr[k]; R[k][1] = minr[k]; R[k][2] = fnn[k]; }
    rot2quat(R, quat);
}

// MGAdapter.make for one face (rfstudio/model/geosplat.py:426-472): out[part][..]
template <typename T>
__device__ void mgadapter_face(const T P[3][3], const T VN[3][3], T means[6][3], T scales[6][3], T quats[6][4], T normals[6][3])
{
    T e1[3], e2[3], fn[3], fnn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { e1[k] = P[1][k] - P[0][k]; e2[k] = P[2][k] - P[0][k]; }
    cross3(e1, e2, fn);
    const T area = Num<T>::clamp_min(norm3(fn), 1e-10f) / 2.0f;
    safe_normalize3(fn, fnn);
    const float u_c[2] = { 1.0f / 9.0f - 1.0f / 24.0f, 2.0f / 9.0f };
    const float a_c[2] = { 0.25f * (1.0f / 3.0f), (1.0f / 12.0f) * 3.0f };
    const float s_r[2] = { 0.5f, 1.3f };
#pragma unroll
    for (int ring = 0; ring < 2; ++ring) {
        const float u = u_c[ring];
        T U[3][3], Nn[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = (i + 1) % 3, l = (i + 2) % 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                U[i][k] = P[i][k] * (1.0f - 2.0f * u) + (P[j][k] + P[l][k]) * u;
                Nn[i][k] = VN[i][k] * (1.0f - 2.0f * u) + (VN[j][k] + VN[l][k]) * u;
            }
        }
        const T a = area * a_c[ring];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int part = ring * 3 + e, i0 = e, i1 = (e + 1) % 3;
            bary2gs(U[i0], U[i1], a, fnn, s_r[ring], means[part], scales[part], quats[part]);
            T nm[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) nm[k] = (Nn[i0][k] + Nn[i1][k]) / 2.0f;
            safe_normalize3(nm, normals[part]);
        }
    }
}

__global__ void __launch_bounds__(256)
mgadapter_fwd_kernel(int F, const float* __restrict__ vertices, const int64_t* __restrict__ faces,
                     const float* __restrict__ vnormals, float* __restrict__ means, float* __restrict__ scales,
                     float* __restrict__ quats, float* __restrict__ normals)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float P[3][3], VN[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int64_t vi = faces[3 * (size_t)f + i];
#pragma unroll
        for (int k = 0; k < 3; ++k) { P[i][k] = vertices[3 * vi + k]; VN[i][k] = vnormals[3 * vi + k]; }
    }
    float m[6][3], s[6][3], q[6][4], n[6][3];
    mgadapter_face<float>(P, VN, m, s, q, n);
#pragma unroll
    for (int part = 0; part < 6; ++part) {
        const size_t row = (size_t)part * F + f;
#pragma unroll
        for (int k = 0; k < 3; ++k) { means[3 * row + k] = m[part][k]; scales[3 * row + k] = s[part][k]; normals[3 * row + k] = n[part][k]; }
        *reinterpret_cast<float4*>(quats + 4 * row) = make_float4(q[part][0], q[part][1], q[part][2], q[part][3]);
    }
}

__global__ void __launch_bounds__(256)
mgadapter_bwd_kernel(int F, const float* __restrict__ vertices, const int64_t* __restrict__ faces,
                     const float* __restrict__ vnormals, const float* __restrict__ v_means,
                     const float* __restrict__ v_scales, const float* __restrict__ v_quats,
                     const float* __restrict__ v_normals, float* __restrict__ v_vertices, float* __restrict__ v_vnormals)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float P[3][3], VN[3][3];
    int64_t vid[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        vid[i] = faces[3 * (size_t)f + i];
#pragma unroll
        for (int k = 0; k < 3; ++k) { P[i][k] = vertices[3 * vid[i] + k]; VN[i][k] = vnormals[3 * vid[i] + k]; }
    }
    // incoming gradients of this face's 6 Gaussians
    float gm[6][3], gs_[6][3], gq[6][4], gn[6][3];
#pragma unroll
    for (int part = 0; part < 6; ++part) {
        const size_t row = (size_t)part * F + f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gm[part][k] = v_means[3 * row + k]; gs_[part][k] = v_scales[3 * row + k];
            gn[part][k] = v_normals ? v_normals[3 * row + k] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) gq[part][k] = v_quats[4 * row + k];
    }
    // forward-mode sweep: one tangent direction per input coordinate
    for (int dir = 0; dir < 18; ++dir) {
        if (dir >= 9 && !v_normals) break;              // shading normals carry no gradient: vnormals untouched
        Dual DP[3][3], DN[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                DP[i][k] = mk(P[i][k], (dir == i * 3 + k) ? 1.0f : 0.0f);
                DN[i][k] = mk(VN[i][k], (dir == 9 + i * 3 + k) ? 1.0f : 0.0f);
            }
        Dual m[6][3], s[6][3], q[6][4], n[6][3];
        mgadapter_face<Dual>(DP, DN, m, s, q, n);
        float acc = 0.0f;
#pragma unroll
        for (int part = 0; part < 6; ++part) {
#pragma unroll
            for (int k = 0; k < 3; ++k) acc += gm[part][k] * m[part][k].d + gs_[part][k] * s[part][k].d + gn[part][k] * n[part][k].d;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += gq[part][k] * q[part][k].d;
        }
        const int which = dir % 9, i = which / 3, k = which % 3;
        float* dst = (dir < 9 ? v_vertices : v_vnormals) + 3 * vid[i] + k;
        if (acc != 0.0f) gs_atomic_add(dst, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// compute_vertex_normals_(fix=True) (rfstudio/graphics/_mesh/_triangle_mesh.py:588-613): area-weighted face normals
// scatter-added to the three corners, then normalised; |n| <= 1e-10 -> (0,0,1).
__global__ void __launch_bounds__(256)
vnormal_scatter_kernel(int F, const float* __restrict__ vertices, const int64_t* __restrict__ faces, float* __restrict__ raw)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    int64_t vid[3]; float P[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        vid[i] = faces[3 * (size_t)f + i];
#pragma unroll
        for (int k = 0; k < 3; ++k) P[i][k] = vertices[3 * vid[i] + k];
    }
    float e1[3], e2[3], fn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { e1[k] = P[1][k] - P[0][k]; e2[k] = P[2][k] - P[0][k]; }
    cross3(e1, e2, fn);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) gs_atomic_add(raw + 3 * vid[i] + k, fn[k]);
}

__global__ void __launch_bounds__(256)
vnormal_normalize_kernel(int V, const float* __restrict__ raw, float* __restrict__ vnormals)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float x = raw[3 * v], y = raw[3 * v + 1], z = raw[3 * v + 2];
    const float l = sqrtf(x * x + y * y + z * z);
    const bool ok = l > 1e-10f;
    const float lc = fmaxf(l, 1e-10f);
    vnormals[3 * v] = ok ? x / lc : 0.0f; vnormals[3 * v + 1] = ok ? y / lc : 0.0f; vnormals[3 * v + 2] = ok ? z / lc : 1.0f;
}

// v_raw = (g - n (n.g)) / |raw| per vertex (zero where the fixing constant was used)
__global__ void __launch_bounds__(256)
vnormal_bwd_vertex_kernel(int V, const float* __restrict__ raw, const float* __restrict__ v_vnormals, float* __restrict__ v_raw)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float x = raw[3 * v], y = raw[3 * v + 1], z = raw[3 * v + 2];
    const float l = sqrtf(x * x + y * y + z * z);
    float gx = 0.0f, gy = 0.0f, gz = 0.0f;
    if (l > 1e-10f) {
        const float nx = x / l, ny = y / l, nz = z / l;
        const float ax = v_vnormals[3 * v], ay = v_vnormals[3 * v + 1], az = v_vnormals[3 * v + 2];
        const float d = nx * ax + ny * ay + nz * az;
        gx = (ax - nx * d) / l; gy = (ay - ny * d) / l; gz = (az - nz * d) / l;
    }
    v_raw[3 * v] = gx; v_raw[3 * v + 1] = gy; v_raw[3 * v + 2] = gz;
}

// per face: g = sum of its corners' v_raw; fn = e1 x e2 -> v_e1 = e2 x g, v_e2 = g x e1
__global__ void __launch_bounds__(256)
vnormal_bwd_face_kernel(int F, const float* __restrict__ vertices, const int64_t* __restrict__ faces,
                        const float* __restrict__ v_raw, float* __restrict__ v_vertices)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    int64_t vid[3]; float P[3][3]; float g[3] = { 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        vid[i] = faces[3 * (size_t)f + i];
#pragma unroll
        for (int k = 0; k < 3; ++k) { P[i][k] = vertices[3 * vid[i] + k]; g[k] += v_raw[3 * vid[i] + k]; }
    }
    float e1[3], e2[3], g1[3], g2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { e1[k] = P[1][k] - P[0][k]; e2[k] = P[2][k] - P[0][k]; }
    cross3(e2, g, g1);
    cross3(g, e1, g2);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gs_atomic_add(v_vertices + 3 * vid[1] + k, g1[k]);
        gs_atomic_add(v_vertices + 3 * vid[2] + k, g2[k]);
        gs_atomic_add(v_vertices + 3 * vid[0] + k, -(g1[k] + g2[k]));
    }
}

extern "C" int gs_mgadapter_fwd(int F, const float* vertices, const int64_t* faces, const float* vnormals,
                                float* means, float* scales, float* quats, float* normals, void* stream)
{
    GS_CHECK_ARG(F >= 0, "bad F");
    if (F == 0) return GS_OK;
    hipLaunchKernelGGL(mgadapter_fwd_kernel, dim3(gs_cdiv(F, 256)), dim3(256), 0, (hipStream_t)stream, F, vertices, faces,
                       vnormals, means, scales, quats, normals);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_mgadapter_bwd(int F, int V, const float* vertices, const int64_t* faces, const float* vnormals,
                                const float* v_means, const float* v_scales, const float* v_quats,
                                const float* v_normals, float* v_vertices, float* v_vnormals, void* stream)
{
    GS_CHECK_ARG(F >= 0 && V >= 0, "bad sizes");
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_HIP(gs_zero_async(v_vertices, sizeof(float) * 3 * (size_t)V, s));
    GS_CHECK_HIP(gs_zero_async(v_vnormals, sizeof(float) * 3 * (size_t)V, s));
    if (F == 0) return GS_OK;
    hipLaunchKernelGGL(mgadapter_bwd_kernel, dim3(gs_cdiv(F, 256)), dim3(256), 0, s, F, vertices, faces, vnormals, v_means,
                       v_scales, v_quats, v_normals, v_vertices, v_vnormals);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_vertex_normals_fwd(int F, int V, const float* vertices, const int64_t* faces, float* raw,
                                     float* vnormals, void* stream)
{
    GS_CHECK_ARG(F >= 0 && V >= 0, "bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (V == 0) return GS_OK;
    GS_CHECK_HIP(gs_zero_async(raw, sizeof(float) * 3 * (size_t)V, s));
    if (F > 0) hipLaunchKernelGGL(vnormal_scatter_kernel, dim3(gs_cdiv(F, 256)), dim3(256), 0, s, F, vertices, faces, raw);
    hipLaunchKernelGGL(vnormal_normalize_kernel, dim3(gs_cdiv(V, 256)), dim3(256), 0, s, V, raw, vnormals);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_vertex_normals_bwd(int F, int V, const float* vertices, const int64_t* faces, const float* raw,
                                     const float* v_vnormals, float* v_raw, float* v_vertices, int accumulate,
                                     void* stream)
{
    GS_CHECK_ARG(F >= 0 && V >= 0, "bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (V == 0) return GS_OK;
    if (!accumulate) GS_CHECK_HIP(gs_zero_async(v_vertices, sizeof(float) * 3 * (size_t)V, s));
    hipLaunchKernelGGL(vnormal_bwd_vertex_kernel, dim3(gs_cdiv(V, 256)), dim3(256), 0, s, V, raw, v_vnormals, v_raw);
    if (F > 0) hipLaunchKernelGGL(vnormal_bwd_face_kernel, dim3(gs_cdiv(F, 256)), dim3(256), 0, s, F, vertices, faces, v_raw, v_vertices);
    GS_CHECK_LAUNCH();
    return GS_OK;
}
