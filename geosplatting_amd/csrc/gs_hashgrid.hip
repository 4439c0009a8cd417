// gs_hashgrid.hip -- SURVEY section 8f rank 3: the multi-resolution hash encoding that produces kd / ks / z per
// Gaussian upstream of the path (rfstudio/model/components/encoding.py:87-241, used at rfstudio/model/geosplat.py:482-520,
// 644-672).  The reference's default backend is tinycudann (CUDA only); what a ROCm user of the reference runs is its
// own `backend='torch'` branch (`pytorch_fwd`, :187-229), and THAT is the semantics restated here (and pinned by
// golden vectors generated from it): every level is hashed (no dense coarse levels), x01 = x/2 + 1/2,
// scaled = x01 * floor(min_res * growth^l), corners = {ceil, floor} per axis, weights = (scaled - floor) for the
// CEIL corner, hash = ((x * 1) xor (y * 2654435761) xor (z * 805459861)) mod T  (+ l * T), products in int64.
//
// Forward: one thread per point loops over the levels (all lanes of a wave are on the same 2^log2_T-entry slice of the
// table at the same time: L2-resident) and writes its L*F contiguous outputs.
// Backward (default, with a workspace): ATOMIC-FREE table gradient -- each workgroup owns a 128 KB LDS slab of one level
// and keeps the contributions of all points that hash into it (hashgrid_bwd_slab_kernel below); the position gradient
// is a separate per-point kernel and flows through the interpolation offsets only (ceil/floor are constant).
// Without a workspace: per-point kernel with fp32 atomics (the F = 2 features of a corner committed by two ADJACENT
// lanes of one instruction = one memory-side request), 2.6x slower at 2 M points.
#include "gs_common.h"

#pragma clang fp contract(off)   // cell indices must round exactly like the reference (x * 0.5 + 0.5, then * scaling)

#define GS_HG_MAX_LEVELS 32

struct HgLevels {
    float scaling[GS_HG_MAX_LEVELS];
};

__device__ __forceinline__ unsigned hg_hash(int x, int y, int z, unsigned mask, unsigned log2_T)
{
    // torch: int32 coords * int64 primes, xor, then python-style modulo by T (a power of two): the low log2_T bits of
    // the two's-complement xor -- identical for negative products
    const long long h = (long long)x ^ ((long long)y * 2654435761ll) ^ ((long long)z * 805459861ll);
    (void)log2_T;
    return (unsigned)((unsigned long long)h & (unsigned long long)mask);
}

template <int F>
__global__ void __launch_bounds__(256)
hashgrid_fwd_kernel(int N, int L, unsigned log2_T, HgLevels lv, const float* __restrict__ x, const float* __restrict__ table,
                    float* __restrict__ out)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const unsigned T = 1u << log2_T, mask = T - 1u;
    const float p[3] = { x[3 * (size_t)n] * 0.5f + 0.5f, x[3 * (size_t)n + 1] * 0.5f + 0.5f, x[3 * (size_t)n + 2] * 0.5f + 0.5f };
    for (int l = 0; l < L; ++l) {
        const float s = lv.scaling[l];
        int c[3], f[3]; float o[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sc = p[k] * s;
            c[k] = (int)ceilf(sc); f[k] = (int)floorf(sc);
            o[k] = sc - (float)f[k];
        }
        const float* tl = table + (size_t)l * T * F;
        float acc[F];
#pragma unroll
        for (int q = 0; q < F; ++q) acc[q] = 0.0f;
        // corner (bx, by, bz): b = 1 -> ceil with weight o, b = 0 -> floor with weight 1 - o; accumulated in the
        // reference's order: x pairs, then y, then z (f_03, f_12, f_56, f_47 -> f0312, f4756 -> value)
        float fx[2][2][F];                                  // [by][bz][feature] after the x interpolation
#pragma unroll
        for (int by = 0; by < 2; ++by)
#pragma unroll
            for (int bz = 0; bz < 2; ++bz) {
                const int yy = by ? c[1] : f[1], zz = bz ? c[2] : f[2];
                const float* pc = tl + (size_t)hg_hash(c[0], yy, zz, mask, log2_T) * F;
                const float* pf = tl + (size_t)hg_hash(f[0], yy, zz, mask, log2_T) * F;
#pragma unroll
                for (int q = 0; q < F; ++q) fx[by][bz][q] = pc[q] * o[0] + pf[q] * (1.0f - o[0]);
            }
#pragma unroll
        for (int q = 0; q < F; ++q) {
            const float z1 = fx[1][1][q] * o[1] + fx[0][1][q] * (1.0f - o[1]);      // bz = 1 (ceil z): f0312
            const float z0 = fx[1][0][q] * o[1] + fx[0][0][q] * (1.0f - o[1]);      // bz = 0:          f4756
            acc[q] = z1 * o[2] + z0 * (1.0f - o[2]);
        }
#pragma unroll
        for (int q = 0; q < F; ++q) out[(size_t)n * L * F + (size_t)l * F + q] = acc[q];
    }
}

template <int F>
__global__ void __launch_bounds__(256)
hashgrid_bwd_kernel(int N, int L, unsigned log2_T, HgLevels lv, const float* __restrict__ x, const float* __restrict__ table,
                    const float* __restrict__ v_out, float table_grad_scale, float* __restrict__ v_table,
                    float* __restrict__ v_x)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = n < N;
    const unsigned T = 1u << log2_T, mask = T - 1u;
    float p[3] = { 0.f, 0.f, 0.f };
    if (live) { p[0] = x[3 * (size_t)n] * 0.5f + 0.5f; p[1] = x[3 * (size_t)n + 1] * 0.5f + 0.5f; p[2] = x[3 * (size_t)n + 2] * 0.5f + 0.5f; }
    float gx[3] = { 0.f, 0.f, 0.f };
    const int lane = threadIdx.x & 63;
    for (int l = 0; l < L; ++l) {
        const float s = lv.scaling[l];
        int c[3], f[3]; float o[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sc = p[k] * s;
            c[k] = (int)ceilf(sc); f[k] = (int)floorf(sc);
            o[k] = sc - (float)f[k];
        }
        float g[F];
#pragma unroll
        for (int q = 0; q < F; ++q) g[q] = live ? v_out[(size_t)n * L * F + (size_t)l * F + q] : 0.0f;
        const float* tl = table + (size_t)l * T * F;
        float* gl = v_table + (size_t)l * T * F;
        float do_[3] = { 0.f, 0.f, 0.f };                      // d value / d o[k], summed over features with g
#pragma unroll
        for (int bx = 0; bx < 2; ++bx)
#pragma unroll
            for (int by = 0; by < 2; ++by)
#pragma unroll
                for (int bz = 0; bz < 2; ++bz) {
                    const unsigned h = hg_hash(bx ? c[0] : f[0], by ? c[1] : f[1], bz ? c[2] : f[2], mask, log2_T);
                    const float wx = bx ? o[0] : 1.0f - o[0], wy = by ? o[1] : 1.0f - o[1], wz = bz ? o[2] : 1.0f - o[2];
                    const float w = wx * wy * wz;
                    float dotg = 0.0f;
                    if (live && v_x) {
#pragma unroll
                        for (int q = 0; q < F; ++q) dotg += tl[(size_t)h * F + q] * g[q];
                    }
                    do_[0] += (bx ? 1.0f : -1.0f) * wy * wz * dotg;
                    do_[1] += (by ? 1.0f : -1.0f) * wx * wz * dotg;
                    do_[2] += (bz ? 1.0f : -1.0f) * wx * wy * dotg;
                    // table gradient: the F features of this corner go out from F adjacent lanes of one instruction
                    // (F = 2: lanes 2j, 2j+1 serve source lane j, then lanes serve source lane 32 + j)
                    float* dst = live ? gl + (size_t)h * F : nullptr;
                    float cv[F];
#pragma unroll
                    for (int q = 0; q < F; ++q) cv[q] = g[q] * w * table_grad_scale;
                    // Points arrive in mesh order, so on the coarse levels most lanes of a wave fall into the same few
                    // cells: while at least a quarter of the remaining lanes share the leader's row, sum them in the
                    // wave and let the leader carry the total (up to 4 rounds); the fine levels skip this after one test.
                    {
                        unsigned long long remaining = __ballot(dst != nullptr);
                        for (int round = 0; round < 4 && remaining != 0ull; ++round) {
                            const int leader = __builtin_ctzll(remaining);
                            const unsigned hl = (unsigned)__builtin_amdgcn_readlane((int)h, leader);
                            const bool same = dst != nullptr && h == hl;
                            const unsigned long long m = __ballot(same);
                            if (__popcll(m) * 4 < __popcll(remaining)) break;
                            float tot[F];
#pragma unroll
                            for (int q = 0; q < F; ++q) tot[q] = gs_wave_sum(same ? cv[q] : 0.0f);
                            if (same) {
                                if (lane == leader) {
#pragma unroll
                                    for (int q = 0; q < F; ++q) cv[q] = tot[q];
                                } else {
                                    dst = nullptr;                       // folded into the leader
                                }
                            }
                            remaining &= ~m;
                        }
                    }
                    const unsigned long long key = (unsigned long long)dst;
#pragma unroll
                    for (int half = 0; half < F; ++half) {
                        const int src = (64 * half + lane) / F, ch = (64 * half + lane) - F * src;
                        const int sa = src << 2;
                        const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(sa, (int)(unsigned)key);
                        const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(sa, (int)(unsigned)(key >> 32));
                        float val = 0.0f;
#pragma unroll
                        for (int q = 0; q < F; ++q) {
                            const float t = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sa, __builtin_bit_cast(int, cv[q])));
                            val = (ch == q) ? t : val;
                        }
                        float* d2 = (float*)(((unsigned long long)hi << 32) | lo);
                        if (d2 != nullptr) gs_atomic_add(d2 + ch, val);     // (8 XCD-private copies with workgroup scope: 12.8 vs 14.8 ms)
                    }
                }
#pragma unroll
        for (int k = 0; k < 3; ++k) gx[k] += do_[k] * s * 0.5f;             // o = x01 * s - floor, x01 = x / 2 + 1/2
    }
    if (live && v_x) { v_x[3 * (size_t)n] = gx[0]; v_x[3 * (size_t)n + 1] = gx[1]; v_x[3 * (size_t)n + 2] = gx[2]; }
}

static int hg_check(int N, int L, int F, int log2_T)
{
    GS_CHECK_ARG(N >= 0, "bad N");
    GS_CHECK_ARG(L >= 1 && L <= GS_HG_MAX_LEVELS, "1 <= num_levels <= 32");
    GS_CHECK_ARG(F == 2, "features_per_level = 2 is built (rfstudio/model/geosplat.py:485-518)");
    GS_CHECK_ARG(log2_T >= 1 && log2_T <= 28, "bad log2_hashmap_size");
    return GS_OK;
}

extern "C" int gs_hashgrid_fwd(int N, int L, int F, int log2_T, const float* scalings_host, const float* x,
                               const float* table, float* out, void* stream)
{
    const int rc = hg_check(N, L, F, log2_T);
    if (rc != GS_OK) return rc;
    if (N == 0) return GS_OK;
    HgLevels lv;
    for (int l = 0; l < GS_HG_MAX_LEVELS; ++l) lv.scaling[l] = l < L ? scalings_host[l] : 0.0f;
    hipLaunchKernelGGL(hashgrid_fwd_kernel<2>, dim3(gs_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, N, L, (unsigned)log2_T, lv,
                       x, table, out);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Atomic-free table gradient (the default): fp32 atomics of this GPU are executed at the memory side (~15 G requests/s,
// whatever their scope), and the hash makes 8 * L * N of them -- 12.8 ms for 2 M points.  Instead every workgroup OWNS
// one 128 KB slab of one level's table (16384 rows x 2 features) in LDS, walks ALL points, recomputes the 8 corner
// hashes and keeps the contributions that fall into its slab (ds_add_f32), then writes the slab with plain stores:
// 16x redundant hashing (VALU is idle anyway) buys the removal of every global atomic.  L * (T / 16384) workgroups =
// 256 at the GaussianField configuration, one per CU.  v_out is first transposed to level-major so that a workgroup
// streams 8 contiguous bytes per point.
#define GS_HG_SLAB_ROWS 16384
#ifndef GS_HG_EXP
#define GS_HG_EXP 0          // timing experiments only (scripts/hashgrid_experiment.py): 1 no LDS atomics, 2 no loads, 4 racy adds
#endif

__global__ void __launch_bounds__(256)
hashgrid_transpose_kernel(int N, int LF, const float* __restrict__ v_out, float* __restrict__ v_lm)
{
    // v_out [N][L*2] -> v_lm [L][N][2]; one thread per (point, level)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int L = LF / 2;
    if (i >= (size_t)N * L) return;
    const size_t n = i / L; const int l = (int)(i - n * L);
    const float2 v = *reinterpret_cast<const float2*>(v_out + n * LF + 2 * l);
    *reinterpret_cast<float2*>(v_lm + ((size_t)l * N + n) * 2) = v;
}

__global__ void __launch_bounds__(1024)
hashgrid_bwd_slab_kernel(int N, int L, unsigned log2_T, HgLevels lv, int slabs_per_level, int parts, const float* __restrict__ x,
                         const float* __restrict__ v_lm, float table_grad_scale, float* __restrict__ v_table, int accumulate)
{
    extern __shared__ float slab[];                                  // [rows][2]
    const unsigned T = 1u << log2_T, mask = T - 1u;
    const int rows = (int)min((unsigned)GS_HG_SLAB_ROWS, T);
    const int b = blockIdx.x;
    const int part = b % parts, sl = (b / parts) % slabs_per_level, l = b / (parts * slabs_per_level);
    for (int i = threadIdx.x; i < rows * 2; i += blockDim.x) slab[i] = 0.0f;
    __syncthreads();
    const float s = lv.scaling[l];
    const float* g_l = v_lm + (size_t)l * N * 2;
    const unsigned row0 = (unsigned)sl * (unsigned)rows;
    const int n_lo = (int)((long long)N * part / parts), n_hi = (int)((long long)N * (part + 1) / parts);
    // PU points per thread and trip, their loads issued together and branch-free (clamped index, gradient zeroed past the
    // end) instead of two dependent L2 round trips per point (gradient, then -- behind the zero-gradient test -- the
    // position).  Measured 5.6 -> 5.4 ms at 2 M points.  Where the time goes (scripts/hashgrid_experiment.py, 2 M points,
    // table gradient only): 4.08 ms as is; 1.87 ms without the LDS atomics; 1.60 ms without atomics and loads (hashing +
    // loop); 2.36 ms with a racy ds_read / add / ds_write in place of ds_add_f32 -- the float LDS atomic itself is the
    // expensive instruction (54 % of the kernel), four times a plain read-modify-write.  Hashing a quarter of the points
    // (binned index queues) and a DPP segmented scan that commits one lane per run of equal rows were both built and
    // both lost.
    constexpr int PU = 4;
    for (int n0 = n_lo + (int)threadIdx.x; n0 < n_hi; n0 += (int)blockDim.x * PU) {
        float2 gq[PU]; float px[PU][3];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int n = n0 + u * (int)blockDim.x;
            const int nn = min(n, n_hi - 1);
#if GS_HG_EXP & 2
            gq[u] = make_float2(1.0f + (float)(nn & 7), 2.0f);           /* timing experiment: no loads */
            px[u][0] = (float)(nn & 1023) * (1.0f / 1024.0f) - 0.3f; px[u][1] = (float)((nn >> 10) & 1023) * (1.0f / 1024.0f) - 0.4f; px[u][2] = 0.1f;
#else
            gq[u] = *reinterpret_cast<const float2*>(g_l + 2 * (size_t)nn);
            px[u][0] = x[3 * (size_t)nn]; px[u][1] = x[3 * (size_t)nn + 1]; px[u][2] = x[3 * (size_t)nn + 2];
#endif
            if (n >= n_hi) gq[u] = make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const float2 g = gq[u];
            if (g.x == 0.0f && g.y == 0.0f) continue;
            int c[3], f[3]; float o[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float sc = (px[u][k] * 0.5f + 0.5f) * s;
                c[k] = (int)ceilf(sc); f[k] = (int)floorf(sc);
                o[k] = sc - (float)f[k];
            }
#pragma unroll
            for (int bx = 0; bx < 2; ++bx)
#pragma unroll
                for (int by = 0; by < 2; ++by)
#pragma unroll
                    for (int bz = 0; bz < 2; ++bz) {
                        const unsigned h = hg_hash(bx ? c[0] : f[0], by ? c[1] : f[1], bz ? c[2] : f[2], mask, log2_T);
                        const unsigned r = h - row0;
                        if (r < (unsigned)rows) {
                            const float w = (bx ? o[0] : 1.0f - o[0]) * (by ? o[1] : 1.0f - o[1]) * (bz ? o[2] : 1.0f - o[2]) * table_grad_scale;
#if GS_HG_EXP & 1
                            if (w == 123.456f) slab[2 * r] = g.x;       /* timing experiment: no LDS atomics */
#elif GS_HG_EXP & 4
                            slab[2 * r] += g.x * w; slab[2 * r + 1] += g.y * w;   /* timing experiment: racy read-modify-write */
#else
                            atomicAdd(&slab[2 * r], g.x * w);            // (wave pre-aggregation of equal rows was measured: 20 % slower)
                            atomicAdd(&slab[2 * r + 1], g.y * w);
#endif
                        }
                    }
        }
    }
    __syncthreads();
    float* dst = v_table + ((size_t)l * T + row0) * 2;
    if (parts == 1) {
        for (int i = threadIdx.x; i < rows * 2; i += blockDim.x) dst[i] = accumulate ? dst[i] + slab[i] : slab[i];
    } else {                                                         // small tables only: several workgroups share a slab
        for (int i = threadIdx.x; i < rows * 2; i += blockDim.x) if (slab[i] != 0.0f) gs_atomic_add(dst + i, slab[i]);
    }
}

// position gradient only (no table gradient): one thread per point
template <int F>
__global__ void __launch_bounds__(256)
hashgrid_bwd_x_kernel(int N, int L, unsigned log2_T, HgLevels lv, const float* __restrict__ x, const float* __restrict__ table,
                      const float* __restrict__ v_out, float* __restrict__ v_x)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const unsigned T = 1u << log2_T, mask = T - 1u;
    const float p[3] = { x[3 * (size_t)n] * 0.5f + 0.5f, x[3 * (size_t)n + 1] * 0.5f + 0.5f, x[3 * (size_t)n + 2] * 0.5f + 0.5f };
    float gx[3] = { 0.f, 0.f, 0.f };
    for (int l = 0; l < L; ++l) {
        const float s = lv.scaling[l];
        int c[3], f[3]; float o[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float sc = p[k] * s;
            c[k] = (int)ceilf(sc); f[k] = (int)floorf(sc);
            o[k] = sc - (float)f[k];
        }
        float g[F];
#pragma unroll
        for (int q = 0; q < F; ++q) g[q] = v_out[(size_t)n * L * F + (size_t)l * F + q];
        const float* tl = table + (size_t)l * T * F;
        float do_[3] = { 0.f, 0.f, 0.f };
#pragma unroll
        for (int bx = 0; bx < 2; ++bx)
#pragma unroll
            for (int by = 0; by < 2; ++by)
#pragma unroll
                for (int bz = 0; bz < 2; ++bz) {
                    const unsigned h = hg_hash(bx ? c[0] : f[0], by ? c[1] : f[1], bz ? c[2] : f[2], mask, log2_T);
                    const float wx = bx ? o[0] : 1.0f - o[0], wy = by ? o[1] : 1.0f - o[1], wz = bz ? o[2] : 1.0f - o[2];
                    float dotg = 0.0f;
#pragma unroll
                    for (int q = 0; q < F; ++q) dotg += tl[(size_t)h * F + q] * g[q];
                    do_[0] += (bx ? 1.0f : -1.0f) * wy * wz * dotg;
                    do_[1] += (by ? 1.0f : -1.0f) * wx * wz * dotg;
                    do_[2] += (bz ? 1.0f : -1.0f) * wx * wy * dotg;
                }
#pragma unroll
        for (int k = 0; k < 3; ++k) gx[k] += do_[k] * s * 0.5f;
    }
    v_x[3 * (size_t)n] = gx[0]; v_x[3 * (size_t)n + 1] = gx[1]; v_x[3 * (size_t)n + 2] = gx[2];
}

extern "C" size_t gs_hashgrid_bwd_ws_bytes(int N, int L, int F)
{
    return sizeof(float) * (size_t)(N > 0 ? N : 1) * (size_t)L * (size_t)F;     // level-major copy of v_out
}

extern "C" int gs_hashgrid_bwd(int N, int L, int F, int log2_T, const float* scalings_host, const float* x,
                               const float* table, const float* v_out, float table_grad_scale, float* v_table,
                               int accumulate, float* v_x, void* ws, size_t ws_bytes, void* stream)
{
    const int rc = hg_check(N, L, F, log2_T);
    if (rc != GS_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)L * ((size_t)1 << log2_T) * F;
    HgLevels lv;
    for (int l = 0; l < GS_HG_MAX_LEVELS; ++l) lv.scaling[l] = l < L ? scalings_host[l] : 0.0f;
    const bool slabs = ws != nullptr && ws_bytes >= gs_hashgrid_bwd_ws_bytes(N, L, F);
    if (N == 0) {
        if (!accumulate) GS_CHECK_HIP(gs_zero_async(v_table, sizeof(float) * n, s));
        return GS_OK;
    }
    if (!slabs) {
        // fallback without workspace: per-point kernel with memory-side atomics (4-5x slower at 2 M points)
        if (!accumulate) GS_CHECK_HIP(gs_zero_async(v_table, sizeof(float) * n, s));
        hipLaunchKernelGGL(hashgrid_bwd_kernel<2>, dim3(gs_cdiv(N, 256)), dim3(256), 0, s, N, L, (unsigned)log2_T, lv, x, table,
                           v_out, table_grad_scale, v_table, v_x);
        GS_CHECK_LAUNCH();
        return GS_OK;
    }
    float* v_lm = (float*)ws;
    hipLaunchKernelGGL(hashgrid_transpose_kernel, dim3(gs_cdiv((int64_t)N * L, 256)), dim3(256), 0, s, N, L * F, v_out, v_lm);
    const unsigned T = 1u << log2_T;
    const int rows = (int)(T < GS_HG_SLAB_ROWS ? T : GS_HG_SLAB_ROWS);
    const int slabs_per_level = (int)(T / (unsigned)rows);
    int parts = 1;
    while (L * slabs_per_level * parts < 192 && parts < 64) parts *= 2;    // small tables: split the points to fill the CUs
    if (parts > 1 && !accumulate) GS_CHECK_HIP(gs_zero_async(v_table, sizeof(float) * n, s));
    const size_t lds = sizeof(float) * 2 * (size_t)rows;
    GS_CHECK_HIP(hipFuncSetAttribute((const void*)hashgrid_bwd_slab_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(hashgrid_bwd_slab_kernel, dim3(L * slabs_per_level * parts), dim3(1024), lds, s, N, L, (unsigned)log2_T, lv,
                       slabs_per_level, parts, x, v_lm, table_grad_scale, v_table, accumulate);
    if (v_x) hipLaunchKernelGGL(hashgrid_bwd_x_kernel<2>, dim3(gs_cdiv(N, 256)), dim3(256), 0, s, N, L, (unsigned)log2_T, lv, x, table,
                                v_out, v_x);
    GS_CHECK_LAUNCH();
    return GS_OK;
}


// ---- table gradient, fixed-point variant ------------------------------------------------------------------------------------
// scripts/micro/lds_atomic_microbench.hip: ds_add_f32 retires ~1 LANE per 3 cycles (a dense wave instruction: ~190
// cycles), ds_add_u32 / ds_add_u64 run at 0.10 / 0.17 cycles per lane -- 18-30x faster; the float slab kernel above spends
// 54 % of its time in that one instruction.  So the slab accumulates in 64-bit FIXED POINT: value * 2^e with one power of
// two per LEVEL, chosen from that level's max|v_out| so that no row sum can overflow (|sum| <= 8 N max scale < 2^62);
// an addend 2^-14 of the largest one still keeps 24 bits.  Integer sums do not depend on the order: the table gradient
// becomes bit-reproducible.  The 8-byte accumulators halve the slab (8192 rows), i.e. double the slabs per level -- paid
// for by binning the points first: one pass appends every point index to the queues of the (level, slab) pairs its 8
// corner rows fall into (~4 of 32: x-neighbours share the upper hash bits), workgroup-aggregated (LDS counters, one
// global atomic per workgroup and queue), and a slab workgroup walks only its queue, with dense lanes.
constexpr int HG_FX_ROWS = 8192;
constexpr int HG_BIN_BLOCK = 1024;

__global__ void __launch_bounds__(HG_BIN_BLOCK)
hashgrid_bin_kernel(int N, int L, unsigned log2_T, HgLevels lv, int slabs, int rows_shift, const float* __restrict__ x,
                    int* __restrict__ q_counts, int* __restrict__ queues)
{
    extern __shared__ int bin_lds[];                 // [L*slabs] block counts -> block bases, then [waves][L*slabs] wave offsets
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LS = L * slabs;
    int* s_cnt = bin_lds;
    int* s_woff = bin_lds + LS;
    const int n = blockIdx.x * HG_BIN_BLOCK + tid;
    const bool valid = n < N;
    const unsigned mask = (1u << log2_T) - 1u;
    for (int i = tid; i < LS; i += HG_BIN_BLOCK) s_cnt[i] = 0;
    __syncthreads();
    float p[3] = {0.f, 0.f, 0.f};
    if (valid) { p[0] = x[3 * (size_t)n] * 0.5f + 0.5f; p[1] = x[3 * (size_t)n + 1] * 0.5f + 0.5f; p[2] = x[3 * (size_t)n + 2] * 0.5f + 0.5f; }
    for (int l = 0; l < L; ++l) {                                     // phase 1: counts (the masks are recomputed in phase 2)
        unsigned long long m = 0ull;
        if (valid) {
            const float sc_ = lv.scaling[l];
            int c[3], f[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { const float v = p[k] * sc_; c[k] = (int)ceilf(v); f[k] = (int)floorf(v); }
#pragma unroll
            for (int b = 0; b < 8; ++b)
                m |= 1ull << (hg_hash(b & 4 ? c[0] : f[0], b & 2 ? c[1] : f[1], b & 1 ? c[2] : f[2], mask, log2_T) >> rows_shift);
        }
        for (int sl = 0; sl < slabs; ++sl) {
            const unsigned long long bal = __ballot((m >> sl) & 1ull);
            if (bal == 0ull) continue;
            if (lane == 0) s_woff[wave * LS + l * slabs + sl] = atomicAdd(&s_cnt[l * slabs + sl], __popcll(bal));
        }
    }
    __syncthreads();
    for (int i = tid; i < LS; i += HG_BIN_BLOCK) {                    // one global atomic per (level, slab) and workgroup
        const int c = s_cnt[i];
        s_cnt[i] = c > 0 ? atomicAdd(&q_counts[i], c) : 0;
    }
    __syncthreads();
    for (int l = 0; l < L; ++l) {                                     // phase 2: write the indices
        unsigned long long m = 0ull;
        if (valid) {
            const float sc_ = lv.scaling[l];
            int c[3], f[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { const float v = p[k] * sc_; c[k] = (int)ceilf(v); f[k] = (int)floorf(v); }
#pragma unroll
            for (int b = 0; b < 8; ++b)
                m |= 1ull << (hg_hash(b & 4 ? c[0] : f[0], b & 2 ? c[1] : f[1], b & 1 ? c[2] : f[2], mask, log2_T) >> rows_shift);
        }
        for (int sl = 0; sl < slabs; ++sl) {
            const bool on = (m >> sl) & 1ull;
            const unsigned long long bal = __ballot(on);
            if (bal == 0ull) continue;
            const int i = l * slabs + sl;
            if (on) queues[(size_t)i * N + s_cnt[i] + s_woff[wave * LS + i] + __popcll(bal & ((1ull << lane) - 1ull))] = n;
        }
    }
}

// per-level max |v| over the level-major copy (bit patterns of non-negative floats order like unsigned integers)
__global__ void __launch_bounds__(1024)
hashgrid_level_max_kernel(int N, const float* __restrict__ v_lm, unsigned* __restrict__ gmax_bits)
{
    __shared__ unsigned red[16];
    const int l = blockIdx.y;
    const float* p = v_lm + (size_t)l * N * 2;
    unsigned m = 0u;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < (size_t)N * 2; i += (size_t)gridDim.x * 1024)
        m = max(m, (unsigned)__float_as_int(fabsf(p[i])));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = max(m, red[w]);
        atomicMax(&gmax_bits[l], m);
    }
}

// e such that 8 N gmax |tgs| 2^e < 2^62 (gmax > 0)
__device__ __forceinline__ int hg_fx_exponent(int N, float gmax, float tgs)
{
    int ex;
    frexpf(8.0f * (float)N * gmax * fabsf(tgs), &ex);               // value < 2^ex
    return 62 - ex;
}

__global__ void __launch_bounds__(1024)
hashgrid_bwd_slabfx_kernel(int N, int L, unsigned log2_T, HgLevels lv, int slabs, const float* __restrict__ x,
                           const float* __restrict__ v_lm, const unsigned* __restrict__ gmax_bits, float table_grad_scale,
                           float* __restrict__ v_table, int accumulate, const int* __restrict__ q_counts,
                           const int* __restrict__ queues)
{
    extern __shared__ unsigned long long fx[];                       // [rows][2] fixed point, two's complement
    const unsigned T = 1u << log2_T, mask = T - 1u;
    const int b = blockIdx.x, sl = b % slabs, l = b / slabs;
    for (int i = threadIdx.x; i < HG_FX_ROWS * 2; i += blockDim.x) fx[i] = 0ull;
    __syncthreads();
    const float gmax = __int_as_float((int)gmax_bits[l]);            // per level: the levels' gradients may differ by decades
    const unsigned row0 = (unsigned)sl * (unsigned)HG_FX_ROWS;
    float* dst = v_table + ((size_t)l * T + row0) * 2;
    if (!(gmax > 0.0f) || !(gmax < 3.0e38f)) {                       // all-zero level: nothing to add; NaN / inf upstream: propagate it
        const bool bad = !(gmax == 0.0f);
        for (int i = threadIdx.x; i < HG_FX_ROWS * 2; i += blockDim.x) {
            if (bad) dst[i] = __int_as_float(0x7fc00000);
            else if (!accumulate) dst[i] = 0.0f;
        }
        return;
    }
    const int e = hg_fx_exponent(N, gmax, table_grad_scale);
    const double up = ldexp(1.0, e), down = ldexp(1.0, -e);
    const float s = lv.scaling[l];
    const float* g_l = v_lm + (size_t)l * N * 2;
    const int* q = queues + (size_t)(l * slabs + sl) * N;
    const int total = q_counts[l * slabs + sl];
    constexpr int PU = 4;
    for (int n0 = (int)threadIdx.x; n0 < total; n0 += (int)blockDim.x * PU) {
        float2 gq[PU]; float px[PU][3];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int i = n0 + u * (int)blockDim.x;
            const int nn = q[min(i, total - 1)];
            gq[u] = *reinterpret_cast<const float2*>(g_l + 2 * (size_t)nn);
            px[u][0] = x[3 * (size_t)nn]; px[u][1] = x[3 * (size_t)nn + 1]; px[u][2] = x[3 * (size_t)nn + 2];
            if (i >= total) gq[u] = make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const float2 g = gq[u];
            if (g.x == 0.0f && g.y == 0.0f) continue;
            int c[3], f[3]; float o[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float sc = (px[u][k] * 0.5f + 0.5f) * s;
                c[k] = (int)ceilf(sc); f[k] = (int)floorf(sc);
                o[k] = sc - (float)f[k];
            }
#pragma unroll
            for (int bx = 0; bx < 2; ++bx)
#pragma unroll
                for (int by = 0; by < 2; ++by)
#pragma unroll
                    for (int bz = 0; bz < 2; ++bz) {
                        const unsigned h = hg_hash(bx ? c[0] : f[0], by ? c[1] : f[1], bz ? c[2] : f[2], mask, log2_T);
                        const unsigned r = h - row0;
                        if (r < (unsigned)HG_FX_ROWS) {
                            const float w = (bx ? o[0] : 1.0f - o[0]) * (by ? o[1] : 1.0f - o[1]) * (bz ? o[2] : 1.0f - o[2]) * table_grad_scale;
                            atomicAdd(&fx[2 * r], (unsigned long long)(long long)((double)(g.x * w) * up));
                            atomicAdd(&fx[2 * r + 1], (unsigned long long)(long long)((double)(g.y * w) * up));
                        }
                    }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HG_FX_ROWS * 2; i += blockDim.x) {
        const float v = (float)((double)(long long)fx[i] * down);
        dst[i] = accumulate ? dst[i] + v : v;
    }
}

extern "C" size_t gs_hashgrid_bwd_fixed_ws_bytes(int N, int L, int F, int log2_T)
{
    const size_t T = (size_t)1 << log2_T;
    if (T < (size_t)HG_FX_ROWS * 4 || T > (size_t)HG_FX_ROWS * 64) return 0;       // 4..64 slabs per level
    const size_t slabs = T / HG_FX_ROWS, n = (size_t)(N > 0 ? N : 1);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return al(gs_hashgrid_bwd_ws_bytes(N, L, F)) + al(sizeof(int) * L * slabs) + 256 + sizeof(int) * L * slabs * n;
}

// Table gradient in 64-bit fixed point over binned points (deterministic); v_x as gs_hashgrid_bwd.  gmax: DEVICE pointer to
// (the per-level maxima of |v_out| that scale the fixed point are reduced on the device, no host sync).  ws: gs_hashgrid_bwd_fixed_ws_bytes (0 = table size unsupported).
extern "C" int gs_hashgrid_bwd_fixed(int N, int L, int F, int log2_T, const float* scalings_host, const float* x,
                                     const float* table, const float* v_out, float table_grad_scale, float* v_table,
                                     int accumulate, float* v_x, void* ws, size_t ws_bytes, void* stream)
{
    const int rc = hg_check(N, L, F, log2_T);
    if (rc != GS_OK) return rc;
    const size_t need = gs_hashgrid_bwd_fixed_ws_bytes(N, L, F, log2_T);
    GS_CHECK_ARG(need > 0, "table size outside the fixed-point path (2^15 .. 2^19 rows per level)");
    GS_CHECK_ARG(ws != nullptr && ws_bytes >= need, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)L * ((size_t)1 << log2_T) * F;
    HgLevels lv;
    for (int l = 0; l < GS_HG_MAX_LEVELS; ++l) lv.scaling[l] = l < L ? scalings_host[l] : 0.0f;
    if (N == 0) {
        if (!accumulate) GS_CHECK_HIP(gs_zero_async(v_table, sizeof(float) * n, s));
        return GS_OK;
    }
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int slabs = (int)(((size_t)1 << log2_T) / HG_FX_ROWS);
    float* v_lm = (float*)ws;
    int* q_counts = (int*)((char*)ws + al(gs_hashgrid_bwd_ws_bytes(N, L, F)));
    unsigned* gmax = (unsigned*)((char*)q_counts + al(sizeof(int) * L * slabs));
    int* queues = (int*)((char*)gmax + 256);
    hipLaunchKernelGGL(hashgrid_transpose_kernel, dim3(gs_cdiv((int64_t)N * L, 256)), dim3(256), 0, s, N, L * F, v_out, v_lm);
    GS_CHECK_HIP(gs_zero_async(q_counts, al(sizeof(int) * L * slabs) + 256, s));            // counters and maxima
    hipLaunchKernelGGL(hashgrid_level_max_kernel, dim3(min(64, gs_cdiv(2 * (int64_t)N, 4096)), L), dim3(1024), 0, s, N, v_lm, gmax);
    int rows_shift = 0;
    while ((1 << rows_shift) < HG_FX_ROWS) ++rows_shift;
    const size_t bin_lds = sizeof(int) * (size_t)L * slabs * (1 + HG_BIN_BLOCK / 64);
    GS_CHECK_ARG(bin_lds <= 160 * 1024, "too many (level, slab) pairs for the binning pass");
    GS_CHECK_HIP(hipFuncSetAttribute((const void*)hashgrid_bin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds));
    hipLaunchKernelGGL(hashgrid_bin_kernel, dim3(gs_cdiv(N, HG_BIN_BLOCK)), dim3(HG_BIN_BLOCK), bin_lds, s, N, L, (unsigned)log2_T, lv,
                       slabs, rows_shift, x, q_counts, queues);
    const size_t lds = sizeof(unsigned long long) * 2 * (size_t)HG_FX_ROWS;
    GS_CHECK_HIP(hipFuncSetAttribute((const void*)hashgrid_bwd_slabfx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(hashgrid_bwd_slabfx_kernel, dim3(L * slabs), dim3(1024), lds, s, N, L, (unsigned)log2_T, lv, slabs, x, v_lm, gmax,
                       table_grad_scale, v_table, accumulate, q_counts, queues);
    if (v_x) hipLaunchKernelGGL(hashgrid_bwd_x_kernel<2>, dim3(gs_cdiv(N, 256)), dim3(256), 0, s, N, L, (unsigned)log2_T, lv, x, table,
                                v_out, v_x);
    GS_CHECK_LAUNCH();
    return GS_OK;
}


// ---- weight gradient of the field's small MLP layers -----------------------------------------------------------------------
// dW[o][i] = sum_n dY[n][o] * X[n][i] with O, I <= 32 and N = millions of Gaussians: a 32 x 32 x N product.  The library GEMM
// behind torch.nn.functional.linear's backward takes 2-5 ms for it at N = 2 M (one 9 ms MLP backward per encoder, against
// 0.7 ms for the forward): skinny in M and N, the K reduction is spread over too few workgroups.  Here every wave streams its
// share of the points once (two rows of dY and X per step = two 256-byte loads) into ONE 32x32 fp32 accumulator tile on
// the matrix unit (v_mfma_f32_32x32x2_f32: exact f32, K = 2 points per instruction), the four waves of a workgroup add
// their tiles through LDS and write a partial; a second kernel sums the partials in a fixed order (deterministic).
// HBM-bound: 8 (O + I) bytes per point.
typedef float gs_f16v __attribute__((ext_vector_type(16)));
constexpr int WG_UNROLL = 8;

__global__ void __launch_bounds__(256) mlp_wgrad_kernel(int64_t N, int O, int I, const float* __restrict__ dY,
                                                        const float* __restrict__ X, float* __restrict__ partial)
{
    __shared__ float red[3][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane >> 5, c = lane & 31;
    const bool ca = c < O, cb = c < I;
    gs_f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int64_t waves = (int64_t)gridDim.x * 4;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    // wave w takes the point pairs w, w + waves, ... in groups of WG_UNROLL pairs
    for (int64_t base = w * 2 * WG_UNROLL; base < N; base += waves * 2 * WG_UNROLL) {
        float a[WG_UNROLL], b[WG_UNROLL];
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) {
            const int64_t n = base + 2 * u + k;
            const int64_t ns = n < N ? n : 0;                       // clamped address, value masked: no branch around the load
            const float av = dY[ns * O + (ca ? c : 0)], bv = X[ns * I + (cb ? c : 0)];
            a[u] = (ca && n < N) ? av : 0.0f;
            b[u] = (cb && n < N) ? bv : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < WG_UNROLL; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    // C/D map: column (i) = lane & 31, row (o) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][((r & 3) + 8 * (r >> 2) + 4 * k) * 32 + c] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = ((r & 3) + 8 * (r >> 2) + 4 * k) * 32 + c;
            partial[(size_t)blockIdx.x * 1024 + e] = ((acc[r] + red[0][e]) + red[1][e]) + red[2][e];
        }
    }
}

// 32 workgroups, each sums 32 of the 1024 tile elements: 32 threads per element walk the partials strided, their 32 sums
// meet in LDS and are added in a fixed order (one workgroup walking all partials serially took 250 us)
__global__ void __launch_bounds__(1024) mlp_wgrad_reduce_kernel(int nb, int O, int I, const float* __restrict__ partial,
                                                                float scale, float* __restrict__ dW, int accumulate)
{
    __shared__ float red[32][33];
    const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + col, o = e >> 5, i = e & 31;
    float s = 0.0f;
    for (int b = part; b < nb; b += 32) s += partial[(size_t)b * 1024 + e];
    red[part][col] = s;
    __syncthreads();
    if (part == 0 && o < O && i < I) {
        float t = 0.0f;
        for (int q = 0; q < 32; ++q) t += red[q][col];
        float* dst = dW + o * I + i;
        *dst = accumulate ? *dst + t * scale : t * scale;
    }
}

static int wgrad_blocks(int64_t N)
{
    const int64_t per_block = 4 * 2 * WG_UNROLL * 8;               // at least 8 trips per wave
    int64_t b = (N + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" size_t gs_mlp_wgrad_ws_bytes(int64_t N) { return (size_t)wgrad_blocks(N) * 1024 * sizeof(float); }

extern "C" int gs_mlp_wgrad(int64_t N, int O, int I, const float* dY, const float* X, float scale, float* dW, int accumulate,
                            void* ws, size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(N >= 0 && O >= 1 && O <= 32 && I >= 1 && I <= 32, "O and I must be in [1, 32]");
    GS_CHECK_ARG(ws_bytes >= gs_mlp_wgrad_ws_bytes(N), "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (!accumulate) GS_CHECK_HIP(gs_zero_async(dW, sizeof(float) * O * I, s));
        return GS_OK;
    }
    const int nb = wgrad_blocks(N);
    hipLaunchKernelGGL(mlp_wgrad_kernel, dim3(nb), dim3(256), 0, s, N, O, I, dY, X, (float*)ws);
    hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3(32), dim3(1024), 0, s, nb, O, I, (const float*)ws, scale, dW, accumulate);
    GS_CHECK_LAUNCH();
    return GS_OK;
}
