// gs_flexicubes.hip -- SURVEY section 8f rank 4: FlexiCubes dual-marching-cubes extraction on a regular grid, the step
// in front of MGAdapter in the stage-1 loop (rfstudio/graphics/_mesh/_flexicubes.py:559-713 with grad_func=None, helpers
// :459-557, :727-802, entropy :715-725; caller rfstudio/model/geosplat.py:751-769).
//
// The reference runs ~60 torch ops per extraction (unique over all cube edges = a sort, stable sort of the quad groups,
// five masked gathers per patch count, index_add / scatter).  Here the grid is implicit and every index the reference
// obtains by sorting is a closed form plus a prefix count:
//   * cube n = i0 + R0 (i1 + R1 i2), grid vertex id = i0 + s1 i1 + s2 i2 (s1 = R0+1, s2 = s1 (R1+1)); corner bit k moves ik;
//   * grid edge = (first vertex f, kind): kind 0 = (f, f-s1) [cube edges 8..11], 1 = (f, f+1) [0,2,4,6], 2 = (f, f+s2)
//     [1,3,5,7]; the reference's unique(dim=0) row order is ascending slot 3f+kind, so ranks are prefix counts over slots;
//   * dual vertex id = class base[#patches of the cube] + #patches * (rank of the cube within its class) + patch;
//   * quad id = rank within (flipped | not flipped) interior surface edges, flipped first; its four cubes in ascending id.
// Three passes: (1) classify + count (two block-scan sweeps, cubes and edge slots), host reads six totals (the
// reference's .item() syncs) and allocates; (2) one thread per surface cube writes its dual vertices and L_dev rows, one
// thread per quad writes the centre vertex and 4 faces; (3) backward: quads push centre gradients to their 4 dual
// vertices / gamma (atomics), then one thread per cube re-derives its patches and pushes to grid vertices / sdf
// (atomics: a grid vertex is shared by up to 8 cubes) and to its own alpha / beta rows (plain stores).
//
// The 256-case patch table and the 36-entry ambiguity table are DERIVED on the host at first use (build_tables): patches
// are the connected components of the occupied corners (of the empty corners for the four body-diagonal tunnel cases),
// ordered by lowest corner, edges ascending; a case is "ambiguous" when its 2 or 3 empty corners checkerboard exactly
// one face.  tests pin the result against meshes the reference itself extracted (tests/golden/ref_flexicubes.npz).
#include <mutex>
#include <string.h>

#include "gs_common.h"

#pragma clang fp contract(off)

namespace {

struct FcTables {
    uint16_t patch_mask[256][4];   // cube-edge bitmask of each patch
    uint32_t owner[256];           // 2 bits per cube edge: which patch holds it
    uint8_t nvd[256];              // number of patches (dual vertices)
    uint8_t kcount[256];           // number of (patch, edge) pairs = crossing edges
    int8_t prob_axis[256];         // 0, or +-(axis+1): checkerboard face towards +-axis
};

__constant__ FcTables c_fc;

constexpr int kEdgeA[12] = {0, 1, 4, 0, 2, 3, 6, 2, 2, 3, 7, 6};   // cube edge e joins corners kEdgeA[e] -> kEdgeB[e] (:18-33)
constexpr int kEdgeB[12] = {1, 5, 5, 4, 3, 7, 7, 6, 0, 1, 5, 4};
// the same, 3 bits per edge, for the device
constexpr uint64_t pack12(const int (&v)[12]) {
    uint64_t r = 0;
    for (int e = 0; e < 12; ++e) r |= (uint64_t)v[e] << (3 * e);
    return r;
}
constexpr uint64_t kPackA = pack12(kEdgeA), kPackB = pack12(kEdgeB);

void build_tables(FcTables& t)
{
    memset(&t, 0, sizeof(t));
    for (int c = 0; c < 256; ++c) {
        const int emp = 255 ^ c;
        int side = c;
        if (__builtin_popcount(emp) == 2) {
            const int lo = __builtin_ctz(emp), hi = 31 - __builtin_clz(emp);
            if ((lo ^ hi) == 7) side = emp;                       // tunnel: two empty corners on a body diagonal
        }
        int left = side, np = 0, kc = 0;
        uint32_t owner = 0;
        while (left) {
            int comp = 1 << __builtin_ctz(left);
            for (bool grow = true; grow;) {
                grow = false;
                for (int e = 0; e < 12; ++e)
                    for (int dir = 0; dir < 2; ++dir) {
                        const int a = dir ? kEdgeB[e] : kEdgeA[e], b = dir ? kEdgeA[e] : kEdgeB[e];
                        if ((comp >> a & 1) && (left >> b & 1) && !(comp >> b & 1)) { comp |= 1 << b; grow = true; }
                    }
            }
            left &= ~comp;
            int mask = 0;
            for (int e = 0; e < 12; ++e)
                if ((comp >> kEdgeA[e] & 1) != (comp >> kEdgeB[e] & 1)) mask |= 1 << e;
            if (mask) {
                t.patch_mask[c][np] = (uint16_t)mask;
                for (int e = 0; e < 12; ++e)
                    if (mask >> e & 1) owner |= (uint32_t)np << (2 * e);
                kc += __builtin_popcount(mask);
                ++np;
            }
        }
        t.nvd[c] = (uint8_t)np; t.kcount[c] = (uint8_t)kc; t.owner[c] = owner;
        const int ne = __builtin_popcount(emp);
        if (ne == 2 || ne == 3) {
            int found = 0, code = 0;
            for (int axis = 0; axis < 3; ++axis)
                for (int val = 0; val < 2; ++val) {
                    int on = 0;
                    for (int k = 0; k < 8; ++k)
                        if ((k >> axis & 1) == val && (emp >> k & 1)) on |= 1 << k;
                    if (__builtin_popcount(on) == 2) {
                        const int lo = __builtin_ctz(on), hi = 31 - __builtin_clz(on);
                        if ((lo ^ hi) == (7 ^ (1 << axis))) { ++found; code = val ? axis + 1 : -(axis + 1); }
                    }
                }
            if (found == 1) t.prob_axis[c] = (int8_t)code;
        }
    }
}

std::once_flag g_tables_once;
hipError_t g_tables_err = hipSuccess;

hipError_t ensure_tables()
{
    std::call_once(g_tables_once, [] {
        static FcTables t;
        build_tables(t);
        g_tables_err = hipMemcpyToSymbol(HIP_SYMBOL(c_fc), &t, sizeof(t));
    });
    return g_tables_err;
}

struct FcGrid {
    int R0, R1, R2, s1, s2, C, Vg;
};

FcGrid make_grid(int R0, int R1, int R2)
{
    FcGrid g;
    g.R0 = R0; g.R1 = R1; g.R2 = R2; g.s1 = R0 + 1; g.s2 = (R0 + 1) * (R1 + 1);
    g.C = R0 * R1 * R2; g.Vg = g.s2 * (R2 + 1);
    return g;
}

__device__ __forceinline__ int fc_origin(const FcGrid& g, int n)
{
    const int x = n % g.R0, y = (n / g.R0) % g.R1, z = n / (g.R0 * g.R1);
    return x + g.s1 * y + g.s2 * z;
}
__device__ __forceinline__ int fc_corner(const FcGrid& g, int origin, int k)
{
    return origin + (k & 1) + ((k >> 1) & 1) * g.s1 + ((k >> 2) & 1) * g.s2;
}
__device__ __forceinline__ int fc_raw_case(const FcGrid& g, const float* __restrict__ sdf, int n)
{
    const int o = fc_origin(g, n);
    int c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) c |= (sdf[fc_corner(g, o, k)] < 0.0f ? 1 : 0) << k;
    return c;
}
// Resolved case of cube n, 0 for cubes the surface does not cross.  _get_case_id (:459-506): an ambiguous cube whose
// neighbour across (d0,d1,d2) -- applied to the C-order unravel of the cube id over (R0,R1,R2), as the reference does
// -- is ambiguous too takes the complement case.
__device__ __forceinline__ int fc_case(const FcGrid& g, const float* __restrict__ sdf, int n)
{
    int raw = fc_raw_case(g, sdf, n);
    if (raw == 0 || raw == 255) return 0;
    const int pa = c_fc.prob_axis[raw];
    if (pa != 0) {
        int u[3] = {n / (g.R1 * g.R2), (n / g.R2) % g.R1, n % g.R2};
        const int dims[3] = {g.R0, g.R1, g.R2};
        const int axis = (pa > 0 ? pa : -pa) - 1;
        u[axis] += pa > 0 ? 1 : -1;
        if (u[axis] >= 0 && u[axis] < dims[axis]) {
            const int raw2 = fc_raw_case(g, sdf, (u[0] * g.R1 + u[1]) * g.R2 + u[2]);
            if (c_fc.prob_axis[raw2] != 0) raw = 255 - raw;
        }
    }
    return raw;
}

// ---- block-wide exclusive scan of NC int components (1024 threads) ----------------------------------------------
constexpr int FC_BLOCK = 1024;
template <int NC>
__device__ __forceinline__ void fc_block_scan(int (&v)[NC], int (&tot)[NC], int* lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int x = v[c];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        incl[c] = x;
        if (lane == 63) lds[wave * NC + c] = x;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int base = 0, all = 0;
        for (int w = 0; w < FC_BLOCK / 64; ++w) {
            const int s = lds[w * NC + c];
            if (w < wave) base += s;
            all += s;
        }
        v[c] = base + incl[c] - v[c];
        tot[c] = all;
    }
    __syncthreads();
}

// cube components: [0..3] cubes with 1..4 patches, [4..7] their (patch, edge) pair counts
constexpr int NCC = 8;
// edge-slot components: flipped quads, unflipped quads, surface edges
constexpr int NCE = 3;

__device__ __forceinline__ void fc_cube_items(const FcGrid& g, const float* __restrict__ sdf, int n, int& cs, int (&v)[NCC])
{
#pragma unroll
    for (int c = 0; c < NCC; ++c) v[c] = 0;
    cs = 0;
    if (n < g.C) {
        cs = fc_case(g, sdf, n);
        const int nv = c_fc.nvd[cs];
        if (nv > 0) { v[nv - 1] = 1; v[4 + nv - 1] = c_fc.kcount[cs]; }
    }
}

// slot -> (exists, crossing, interior, flip)
struct FcSlot {
    bool cross, interior, flip;
};
__device__ __forceinline__ FcSlot fc_slot(const FcGrid& g, const float* __restrict__ sdf, int64_t slot)
{
    FcSlot r = {false, false, false};
    if (slot >= 3 * (int64_t)g.Vg) return r;
    const int f = (int)(slot / 3), kind = (int)(slot - 3 * (int64_t)f);
    const int i0 = f % g.s1, i1 = (f / g.s1) % (g.R1 + 1), i2 = f / g.s2;
    const bool exists = kind == 0 ? i1 >= 1 : (kind == 1 ? i0 < g.R0 : i2 < g.R2);
    if (!exists) return r;
    const int second = kind == 0 ? f - g.s1 : (kind == 1 ? f + 1 : f + g.s2);
    const float sf = sdf[f], ss = sdf[second];
    r.cross = (sf < 0.0f) != (ss < 0.0f);
    if (!r.cross) return r;
    r.interior = kind == 0 ? (i0 >= 1 && i0 < g.R0 && i2 >= 1 && i2 < g.R2)
               : kind == 1 ? (i1 >= 1 && i1 < g.R1 && i2 >= 1 && i2 < g.R2)
                           : (i0 >= 1 && i0 < g.R0 && i1 >= 1 && i1 < g.R1);
    r.flip = sf > 0.0f;                                             // s_edges[:,0] > 0 (:770)
    return r;
}
__device__ __forceinline__ void fc_slot_items(const FcSlot& s, int (&v)[NCE])
{
    v[0] = s.interior && s.flip; v[1] = s.interior && !s.flip; v[2] = s.cross;
}

__global__ void __launch_bounds__(FC_BLOCK) fc_cube_count_kernel(FcGrid g, const float* __restrict__ sdf, int* __restrict__ blk)
{
    __shared__ int lds[(FC_BLOCK / 64) * NCC];
    int cs, v[NCC], tot[NCC];
    fc_cube_items(g, sdf, blockIdx.x * FC_BLOCK + threadIdx.x, cs, v);
    fc_block_scan<NCC>(v, tot, lds);
    if (threadIdx.x < NCC) blk[blockIdx.x * NCC + threadIdx.x] = tot[threadIdx.x];
}

__global__ void __launch_bounds__(FC_BLOCK) fc_slot_count_kernel(FcGrid g, const float* __restrict__ sdf, int* __restrict__ blk)
{
    __shared__ int lds[(FC_BLOCK / 64) * NCE];
    int v[NCE], tot[NCE];
    fc_slot_items(fc_slot(g, sdf, (int64_t)blockIdx.x * FC_BLOCK + threadIdx.x), v);
    fc_block_scan<NCE>(v, tot, lds);
    if (threadIdx.x < NCE) blk[blockIdx.x * NCE + threadIdx.x] = tot[threadIdx.x];
}

// exclusive scan of the per-block totals in place (single workgroup); totals to tot_out[NC]
template <int NC>
__global__ void __launch_bounds__(FC_BLOCK) fc_scan_blocks_kernel(int nb, int* __restrict__ blk, int* __restrict__ tot_out)
{
    __shared__ int lds[(FC_BLOCK / 64) * NC];
    const int per = (nb + FC_BLOCK - 1) / FC_BLOCK;
    const int b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
    int v[NC], tot[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = 0;
    for (int b = b0; b < b1; ++b)
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] += blk[b * NC + c];
    fc_block_scan<NC>(v, tot, lds);
    for (int b = b0; b < b1; ++b)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int t = blk[b * NC + c];
            blk[b * NC + c] = v[c];
            v[c] += t;
        }
    if (threadIdx.x == 0)
#pragma unroll
        for (int c = 0; c < NC; ++c) tot_out[c] = tot[c];
}

// header of the workspace (ints)
enum { H_CUBE_TOT = 0 /*8*/, H_SLOT_TOT = 8 /*3*/, H_INTS = 64 };

__global__ void fc_counts_kernel(const int* __restrict__ hdr, int64_t* __restrict__ counts)
{
    const int* ct = hdr + H_CUBE_TOT; const int* st = hdr + H_SLOT_TOT;
    int64_t N = 0, Q = 0, K = 0;
    for (int j = 0; j < 4; ++j) { N += ct[j]; Q += (int64_t)(j + 1) * ct[j]; K += ct[4 + j]; }
    counts[0] = N; counts[1] = Q; counts[2] = K; counts[3] = st[2]; counts[4] = (int64_t)st[0] + st[1]; counts[5] = st[0];
    counts[6] = 0; counts[7] = 0;
}

__global__ void __launch_bounds__(FC_BLOCK) fc_cube_assign_kernel(FcGrid g, const float* __restrict__ sdf, const int* __restrict__ blk,
                                                                  const int* __restrict__ hdr, uint8_t* __restrict__ case8,
                                                                  int* __restrict__ vd_base, int* __restrict__ k_base)
{
    __shared__ int lds[(FC_BLOCK / 64) * NCC];
    const int n = blockIdx.x * FC_BLOCK + threadIdx.x;
    int cs, v[NCC], tot[NCC];
    fc_cube_items(g, sdf, n, cs, v);
    fc_block_scan<NCC>(v, tot, lds);
    if (n >= g.C) return;
    case8[n] = (uint8_t)cs;
    const int nv = c_fc.nvd[cs];
    if (nv == 0) return;
    const int* ct = hdr + H_CUBE_TOT;
    int vb = 0, kb = 0;
    for (int j = 0; j < nv - 1; ++j) { vb += (j + 1) * ct[j]; kb += ct[4 + j]; }
    vd_base[n] = vb + nv * (blk[blockIdx.x * NCC + nv - 1] + v[nv - 1]);
    k_base[n] = kb + blk[blockIdx.x * NCC + 4 + nv - 1] + v[4 + nv - 1];
}

__global__ void __launch_bounds__(FC_BLOCK) fc_slot_assign_kernel(FcGrid g, const float* __restrict__ sdf, const int* __restrict__ blk,
                                                                  const int* __restrict__ hdr, int* __restrict__ quad_id)
{
    __shared__ int lds[(FC_BLOCK / 64) * NCE];
    const int64_t slot = (int64_t)blockIdx.x * FC_BLOCK + threadIdx.x;
    const FcSlot s = fc_slot(g, sdf, slot);
    int v[NCE], tot[NCE];
    fc_slot_items(s, v);
    fc_block_scan<NCE>(v, tot, lds);
    if (slot >= 3 * (int64_t)g.Vg) return;
    int q = -1;
    if (s.interior) q = s.flip ? blk[blockIdx.x * NCE + 0] + v[0] : hdr[H_SLOT_TOT + 0] + blk[blockIdx.x * NCE + 1] + v[1];
    quad_id[slot] = q;
}

// ---- per-cube forward ------------------------------------------------------------------------------------------------
struct FcParams {
    float weight_scale, sdf_eps;    // sdf_eps < 0: None
};

__device__ __forceinline__ float fc_interp_w(float A, float B, const FcParams& p)
{
    float w = A / (A - B);                                          // _linear_interp (:551-556)
    if (p.sdf_eps >= 0.0f) w = (1.0f - p.sdf_eps) * w + p.sdf_eps / 2.0f;
    return w;
}

__global__ void __launch_bounds__(256) fc_dual_vertex_kernel(FcGrid g, FcParams prm, const float* __restrict__ vertices,
                                                             const float* __restrict__ sdf, const float* __restrict__ alpha,
                                                             const float* __restrict__ beta, const uint8_t* __restrict__ case8,
                                                             const int* __restrict__ vd_base, const int* __restrict__ k_base,
                                                             float* __restrict__ out_vertices, float* __restrict__ L_dev)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= g.C) return;
    const int cs = case8[n];
    if (cs == 0) return;
    const int o = fc_origin(g, n);
    const int nv = c_fc.nvd[cs];
    int kk = k_base[n];
    const int vid = vd_base[n];
    for (int p = 0; p < nv; ++p) {
        const int mask = c_fc.patch_mask[cs][p];
        float S0 = 0.0f, S1 = 0.0f, S2 = 0.0f, B = 0.0f;
        for (int m = mask; m; m &= m - 1) {
            const int e = __builtin_ctz(m);
            const int ca = (int)(kPackA >> (3 * e)) & 7, cb = (int)(kPackB >> (3 * e)) & 7;
            const int va = fc_corner(g, o, ca), vb = fc_corner(g, o, cb);
            float sa = sdf[va], sb = sdf[vb];
            if (alpha) {
                sa = sa * (tanhf(alpha[(size_t)n * 8 + ca]) * prm.weight_scale + 1.0f);
                sb = sb * (tanhf(alpha[(size_t)n * 8 + cb]) * prm.weight_scale + 1.0f);
            }
            const float w = fc_interp_w(sa, sb, prm);
            const float bt = beta ? tanhf(beta[(size_t)n * 12 + e]) * prm.weight_scale + 1.0f : 1.0f;
            S0 += (vertices[3 * (size_t)vb + 0] * w + vertices[3 * (size_t)va + 0] * (1.0f - w)) * bt;
            S1 += (vertices[3 * (size_t)vb + 1] * w + vertices[3 * (size_t)va + 1] * (1.0f - w)) * bt;
            S2 += (vertices[3 * (size_t)vb + 2] * w + vertices[3 * (size_t)va + 2] * (1.0f - w)) * bt;
            B += bt;
        }
        const float v0 = S0 / B, v1 = S1 / B, v2 = S2 / B;
        out_vertices[3 * (size_t)(vid + p) + 0] = v0;
        out_vertices[3 * (size_t)(vid + p) + 1] = v1;
        out_vertices[3 * (size_t)(vid + p) + 2] = v2;
        // L_dev (:727-741): |dist - mean dist| of the plain zero crossings to the dual vertex
        float d[7], sum = 0.0f;
        int cnt = 0;
        for (int m = mask; m; m &= m - 1) {
            const int e = __builtin_ctz(m);
            const int va = fc_corner(g, o, (int)(kPackA >> (3 * e)) & 7), vb = fc_corner(g, o, (int)(kPackB >> (3 * e)) & 7);
            const float w = fc_interp_w(sdf[va], sdf[vb], prm);
            const float z0 = vertices[3 * (size_t)vb + 0] * w + vertices[3 * (size_t)va + 0] * (1.0f - w) - v0;
            const float z1 = vertices[3 * (size_t)vb + 1] * w + vertices[3 * (size_t)va + 1] * (1.0f - w) - v1;
            const float z2 = vertices[3 * (size_t)vb + 2] * w + vertices[3 * (size_t)va + 2] * (1.0f - w) - v2;
            const float dist = sqrtf(z0 * z0 + z1 * z1 + z2 * z2);
            d[cnt++] = dist; sum += dist;
        }
        const float mean = sum / (float)cnt;
        for (int j = 0; j < cnt; ++j) L_dev[kk++] = fabsf(d[j] - mean);
    }
}

// the four cubes around an interior edge (ascending id) and the cube edge each of them sees it as
__device__ __forceinline__ void fc_quad_cubes(const FcGrid& g, int f, int kind, int (&cube)[4], int (&cedge)[4])
{
    const int lo = kind == 0 ? f - g.s1 : f;
    const int x = lo % g.s1, y = (lo / g.s1) % (g.R1 + 1), z = lo / g.s2;
    auto cid = [&](int a, int b, int c) { return a + g.R0 * (b + g.R1 * c); };
    if (kind == 1) {
        cube[0] = cid(x, y - 1, z - 1); cube[1] = cid(x, y, z - 1); cube[2] = cid(x, y - 1, z); cube[3] = cid(x, y, z);
        cedge[0] = 6; cedge[1] = 2; cedge[2] = 4; cedge[3] = 0;
    } else if (kind == 0) {
        cube[0] = cid(x - 1, y, z - 1); cube[1] = cid(x, y, z - 1); cube[2] = cid(x - 1, y, z); cube[3] = cid(x, y, z);
        cedge[0] = 10; cedge[1] = 11; cedge[2] = 9; cedge[3] = 8;
    } else {
        cube[0] = cid(x - 1, y - 1, z); cube[1] = cid(x, y - 1, z); cube[2] = cid(x - 1, y, z); cube[3] = cid(x, y, z);
        cedge[0] = 5; cedge[1] = 7; cedge[2] = 1; cedge[3] = 3;
    }
}

struct FcQuad {
    int vd[4], cube[4];
    float gm[4];
};
// dual vertices of the quad in the reference's winding (:771-772) and their cubes' activated gamma
__device__ __forceinline__ void fc_quad_setup(const FcGrid& g, const FcParams& prm, int f, int kind, bool flip,
                                              const uint8_t* __restrict__ case8, const int* __restrict__ vd_base,
                                              const float* __restrict__ gamma, FcQuad& q)
{
    int cube[4], cedge[4];
    fc_quad_cubes(g, f, kind, cube, cedge);
    const int perm_f[4] = {0, 1, 3, 2}, perm_n[4] = {2, 3, 1, 0};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int s = flip ? perm_f[t] : perm_n[t];
        const int c = cube[s];
        q.cube[t] = c;
        q.vd[t] = vd_base[c] + (int)((c_fc.owner[case8[c]] >> (2 * cedge[s])) & 3u);
        q.gm[t] = gamma ? (1.0f / (1.0f + expf(-gamma[c]))) * prm.weight_scale + (1.0f - prm.weight_scale) / 2.0f : 1.0f;
    }
}

__global__ void __launch_bounds__(256) fc_quad_kernel(FcGrid g, FcParams prm, const float* __restrict__ sdf,
                                                      const float* __restrict__ gamma, const uint8_t* __restrict__ case8,
                                                      const int* __restrict__ vd_base, const int* __restrict__ quad_id, int Q,
                                                      float* __restrict__ out_vertices, int64_t* __restrict__ faces)
{
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= 3 * (int64_t)g.Vg) return;
    const int qi = quad_id[slot];
    if (qi < 0) return;
    const int f = (int)(slot / 3), kind = (int)(slot - 3 * (int64_t)f);
    FcQuad q;
    fc_quad_setup(g, prm, f, kind, sdf[f] > 0.0f, case8, vd_base, gamma, q);
    const float g02 = q.gm[0] * q.gm[2], g13 = q.gm[1] * q.gm[3];
    const float wsum = (g02 + g13) + 1e-8f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v02 = (out_vertices[3 * (size_t)q.vd[0] + c] + out_vertices[3 * (size_t)q.vd[2] + c]) / 2.0f;
        const float v13 = (out_vertices[3 * (size_t)q.vd[1] + c] + out_vertices[3 * (size_t)q.vd[3] + c]) / 2.0f;
        out_vertices[3 * (size_t)(Q + qi) + c] = (v02 * g02 + v13 * g13) / wsum;
    }
    int64_t* fo = faces + 12 * (size_t)qi;
#pragma unroll
    for (int t = 0; t < 4; ++t) { fo[3 * t + 0] = q.vd[t]; fo[3 * t + 1] = q.vd[(t + 1) & 3]; fo[3 * t + 2] = Q + qi; }
}

// ---- backward --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fc_quad_bwd_kernel(FcGrid g, FcParams prm, const float* __restrict__ sdf,
                                                          const float* __restrict__ gamma, const uint8_t* __restrict__ case8,
                                                          const int* __restrict__ vd_base, const int* __restrict__ quad_id, int Q,
                                                          const float* __restrict__ out_vertices, const float* __restrict__ v_out,
                                                          float* __restrict__ g_vd, float* __restrict__ g_gamma)
{
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= 3 * (int64_t)g.Vg) return;
    const int qi = quad_id[slot];
    if (qi < 0) return;
    const int f = (int)(slot / 3), kind = (int)(slot - 3 * (int64_t)f);
    FcQuad q;
    fc_quad_setup(g, prm, f, kind, sdf[f] > 0.0f, case8, vd_base, gamma, q);
    const float g02 = q.gm[0] * q.gm[2], g13 = q.gm[1] * q.gm[3];
    const float wsum = (g02 + g13) + 1e-8f;
    float d02 = 0.0f, d13 = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float go = v_out[3 * (size_t)(Q + qi) + c];
        const float v02 = (out_vertices[3 * (size_t)q.vd[0] + c] + out_vertices[3 * (size_t)q.vd[2] + c]) / 2.0f;
        const float v13 = (out_vertices[3 * (size_t)q.vd[1] + c] + out_vertices[3 * (size_t)q.vd[3] + c]) / 2.0f;
        const float ctr = (v02 * g02 + v13 * g13) / wsum;
        const float a02 = go * g02 / wsum * 0.5f, a13 = go * g13 / wsum * 0.5f;
        gs_atomic_add(&g_vd[3 * (size_t)q.vd[0] + c], a02); gs_atomic_add(&g_vd[3 * (size_t)q.vd[2] + c], a02);
        gs_atomic_add(&g_vd[3 * (size_t)q.vd[1] + c], a13); gs_atomic_add(&g_vd[3 * (size_t)q.vd[3] + c], a13);
        d02 += go * (v02 - ctr) / wsum; d13 += go * (v13 - ctr) / wsum;
    }
    if (g_gamma) {
        const float dg[4] = {d02 * q.gm[2], d13 * q.gm[3], d02 * q.gm[0], d13 * q.gm[1]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float sg = 1.0f / (1.0f + expf(-gamma[q.cube[t]]));
            gs_atomic_add(&g_gamma[q.cube[t]], dg[t] * prm.weight_scale * sg * (1.0f - sg));
        }
    }
}

__global__ void __launch_bounds__(256) fc_dual_vertex_bwd_kernel(FcGrid g, FcParams prm, const float* __restrict__ vertices,
                                                                 const float* __restrict__ sdf, const float* __restrict__ alpha,
                                                                 const float* __restrict__ beta, const uint8_t* __restrict__ case8,
                                                                 const int* __restrict__ vd_base, const int* __restrict__ k_base,
                                                                 const float* __restrict__ out_vertices, const float* __restrict__ g_vd,
                                                                 const float* __restrict__ v_L, float* __restrict__ g_vertices,
                                                                 float* __restrict__ g_sdf, float* __restrict__ g_alpha,
                                                                 float* __restrict__ g_beta)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= g.C) return;
    const int cs = case8[n];
    if (cs == 0) return;
    const int o = fc_origin(g, n);
    const int nv = c_fc.nvd[cs];
    int kk = k_base[n];
    const int vid = vd_base[n];
    float ga[8] = {0, 0, 0, 0, 0, 0, 0, 0};                       // d/d activated alpha of the 8 corners
    for (int p = 0; p < nv; ++p) {
        const int mask = c_fc.patch_mask[cs][p];
        const float v0 = out_vertices[3 * (size_t)(vid + p) + 0], v1 = out_vertices[3 * (size_t)(vid + p) + 1],
                    v2 = out_vertices[3 * (size_t)(vid + p) + 2];
        float gv0 = g_vd[3 * (size_t)(vid + p) + 0], gv1 = g_vd[3 * (size_t)(vid + p) + 1], gv2 = g_vd[3 * (size_t)(vid + p) + 2];
        // pass 1: L_dev -- distances, their mean, the cotangent of every distance
        float d[7], gd[7], sum = 0.0f, B = 0.0f;
        int cnt = 0;
        for (int m = mask; m; m &= m - 1) {
            const int e = __builtin_ctz(m);
            const int va = fc_corner(g, o, (int)(kPackA >> (3 * e)) & 7), vb = fc_corner(g, o, (int)(kPackB >> (3 * e)) & 7);
            const float w = fc_interp_w(sdf[va], sdf[vb], prm);
            const float z0 = vertices[3 * (size_t)vb + 0] * w + vertices[3 * (size_t)va + 0] * (1.0f - w) - v0;
            const float z1 = vertices[3 * (size_t)vb + 1] * w + vertices[3 * (size_t)va + 1] * (1.0f - w) - v1;
            const float z2 = vertices[3 * (size_t)vb + 2] * w + vertices[3 * (size_t)va + 2] * (1.0f - w) - v2;
            d[cnt] = sqrtf(z0 * z0 + z1 * z1 + z2 * z2); sum += d[cnt]; ++cnt;
            B += beta ? tanhf(beta[(size_t)n * 12 + e]) * prm.weight_scale + 1.0f : 1.0f;
        }
        const float mean = sum / (float)cnt;
        float ssum = 0.0f;
        for (int j = 0; j < cnt; ++j) {
            const float df = d[j] - mean;
            gd[j] = v_L ? (df > 0.0f ? v_L[kk + j] : (df < 0.0f ? -v_L[kk + j] : 0.0f)) : 0.0f;
            ssum += gd[j];
        }
        for (int j = 0; j < cnt; ++j) gd[j] -= ssum / (float)cnt;
        kk += cnt;
        // pass 2: the distance cotangents reach the zero crossings and the dual vertex
        float zg[7][3];
        {
            int j = 0;
            for (int m = mask; m; m &= m - 1, ++j) {
                const int e = __builtin_ctz(m);
                const int va = fc_corner(g, o, (int)(kPackA >> (3 * e)) & 7), vb = fc_corner(g, o, (int)(kPackB >> (3 * e)) & 7);
                const float w = fc_interp_w(sdf[va], sdf[vb], prm);
                const float z0 = vertices[3 * (size_t)vb + 0] * w + vertices[3 * (size_t)va + 0] * (1.0f - w) - v0;
                const float z1 = vertices[3 * (size_t)vb + 1] * w + vertices[3 * (size_t)va + 1] * (1.0f - w) - v1;
                const float z2 = vertices[3 * (size_t)vb + 2] * w + vertices[3 * (size_t)va + 2] * (1.0f - w) - v2;
                const float s = d[j] > 0.0f ? gd[j] / d[j] : 0.0f;
                zg[j][0] = s * z0; zg[j][1] = s * z1; zg[j][2] = s * z2;
                gv0 -= zg[j][0]; gv1 -= zg[j][1]; gv2 -= zg[j][2];
            }
        }
        // pass 3: dual vertex = sum(ue * beta) / sum(beta)
        int j = 0;
        for (int m = mask; m; m &= m - 1, ++j) {
            const int e = __builtin_ctz(m);
            const int ca = (int)(kPackA >> (3 * e)) & 7, cb = (int)(kPackB >> (3 * e)) & 7;
            const int va = fc_corner(g, o, ca), vb = fc_corner(g, o, cb);
            const float sa = sdf[va], sb = sdf[vb];
            float aa = 1.0f, ab = 1.0f;
            if (alpha) {
                aa = tanhf(alpha[(size_t)n * 8 + ca]) * prm.weight_scale + 1.0f;
                ab = tanhf(alpha[(size_t)n * 8 + cb]) * prm.weight_scale + 1.0f;
            }
            const float A = sa * aa, Bq = sb * ab;
            const float w = fc_interp_w(A, Bq, prm);
            float tb = 0.0f, bt = 1.0f;
            if (beta) { tb = tanhf(beta[(size_t)n * 12 + e]); bt = tb * prm.weight_scale + 1.0f; }
            const float xa[3] = {vertices[3 * (size_t)va + 0], vertices[3 * (size_t)va + 1], vertices[3 * (size_t)va + 2]};
            const float xb[3] = {vertices[3 * (size_t)vb + 0], vertices[3 * (size_t)vb + 1], vertices[3 * (size_t)vb + 2]};
            const float gv[3] = {gv0, gv1, gv2}, vd[3] = {v0, v1, v2};
            // plain zero crossing (weights wz) and alpha-weighted crossing (weights w)
            const float wz = fc_interp_w(sa, sb, prm);
            float dw = 0.0f, dwz = 0.0f, dbt = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float gue = gv[c] * bt / B;                  // d/d ue
                const float ue = xb[c] * w + xa[c] * (1.0f - w);
                dbt += gv[c] * (ue - vd[c]) / B;
                dw += gue * (xb[c] - xa[c]);
                dwz += zg[j][c] * (xb[c] - xa[c]);
                gs_atomic_add(&g_vertices[3 * (size_t)va + c], gue * (1.0f - w) + zg[j][c] * (1.0f - wz));
                gs_atomic_add(&g_vertices[3 * (size_t)vb + c], gue * w + zg[j][c] * wz);
            }
            if (prm.sdf_eps >= 0.0f) { dw *= 1.0f - prm.sdf_eps; dwz *= 1.0f - prm.sdf_eps; }
            // w = A / (A - Bq)
            const float den = (A - Bq) * (A - Bq), denz = (sa - sb) * (sa - sb);
            const float dA = dw * (-Bq) / den, dBq = dw * A / den;
            gs_atomic_add(&g_sdf[va], dA * aa + dwz * (-sb) / denz);
            gs_atomic_add(&g_sdf[vb], dBq * ab + dwz * sa / denz);
            ga[ca] += dA * sa; ga[cb] += dBq * sb;
            if (g_beta) g_beta[(size_t)n * 12 + e] = dbt * prm.weight_scale * (1.0f - tb * tb);
        }
    }
    if (g_alpha) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float t = tanhf(alpha[(size_t)n * 8 + k]);
            g_alpha[(size_t)n * 8 + k] = ga[k] * prm.weight_scale * (1.0f - t * t);
        }
    }
}

// ---- entropy (:715-725) ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fc_bce(float x, float t) { return fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x))); }

__global__ void __launch_bounds__(256) fc_entropy_kernel(FcGrid g, const float* __restrict__ sdf, double* __restrict__ partial)
{
    __shared__ double red[256];
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    const FcSlot s = fc_slot(g, sdf, slot);
    if (s.cross) {
        const int f = (int)(slot / 3), kind = (int)(slot - 3 * (int64_t)f);
        const int second = kind == 0 ? f - g.s1 : (kind == 1 ? f + 1 : f + g.s2);
        const float a = sdf[f], b = sdf[second];
        acc = (double)fc_bce(a, b > 0.0f ? 1.0f : 0.0f) + (double)fc_bce(b, a > 0.0f ? 1.0f : 0.0f);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) fc_entropy_finalize_kernel(int nb, const double* __restrict__ partial, const int* __restrict__ hdr,
                                                                  float* __restrict__ out)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) acc += partial[b];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)hdr[H_SLOT_TOT + 2]);
}

__global__ void __launch_bounds__(256) fc_entropy_bwd_kernel(FcGrid g, const float* __restrict__ sdf, const int* __restrict__ hdr,
                                                             const float* __restrict__ v_out, float* __restrict__ g_sdf)
{
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const FcSlot s = fc_slot(g, sdf, slot);
    if (!s.cross) return;
    const int f = (int)(slot / 3), kind = (int)(slot - 3 * (int64_t)f);
    const int second = kind == 0 ? f - g.s1 : (kind == 1 ? f + 1 : f + g.s2);
    const float a = sdf[f], b = sdf[second];
    const float k = v_out[0] / (float)hdr[H_SLOT_TOT + 2];
    gs_atomic_add(&g_sdf[f], k * (1.0f / (1.0f + expf(-a)) - (b > 0.0f ? 1.0f : 0.0f)));
    gs_atomic_add(&g_sdf[second], k * (1.0f / (1.0f + expf(-b)) - (a > 0.0f ? 1.0f : 0.0f)));
}

// ---- workspace layout --------------------------------------------------------------------------------------------------
struct FcWs {
    int* hdr; uint8_t* case8; int* vd_base; int* k_base; int* quad_id; int* blk_cube; int* blk_slot; double* partial;
    int nb_cube, nb_slot, nb_ent;
    size_t bytes;
};
size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
FcWs fc_ws(const FcGrid& g, void* base)
{
    FcWs w;
    w.nb_cube = gs_cdiv(g.C, FC_BLOCK); w.nb_slot = gs_cdiv(3 * (int64_t)g.Vg, FC_BLOCK); w.nb_ent = gs_cdiv(3 * (int64_t)g.Vg, 256);
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += al256(n); return (char*)base + o; };
    w.hdr = (int*)take(sizeof(int) * H_INTS);
    w.case8 = (uint8_t*)take((size_t)g.C);
    w.vd_base = (int*)take(sizeof(int) * (size_t)g.C);
    w.k_base = (int*)take(sizeof(int) * (size_t)g.C);
    w.quad_id = (int*)take(sizeof(int) * 3 * (size_t)g.Vg);
    w.blk_cube = (int*)take(sizeof(int) * NCC * (size_t)w.nb_cube);
    w.blk_slot = (int*)take(sizeof(int) * NCE * (size_t)w.nb_slot);
    w.partial = (double*)take(sizeof(double) * (size_t)w.nb_ent);
    w.bytes = off;
    return w;
}

bool grid_ok(int R0, int R1, int R2)
{
    if (R0 < 1 || R1 < 1 || R2 < 1) return false;
    const int64_t vg = (int64_t)(R0 + 1) * (R1 + 1) * (R2 + 1);
    return 3 * vg < ((int64_t)1 << 31);
}

}  // namespace

extern "C" size_t gs_flexicubes_ws_bytes(int R0, int R1, int R2)
{
    if (!grid_ok(R0, R1, R2)) return 0;
    return fc_ws(make_grid(R0, R1, R2), nullptr).bytes;
}

extern "C" int gs_flexicubes_count(int R0, int R1, int R2, const float* sdf, void* ws, size_t ws_bytes, int64_t* counts,
                                   void* stream)
{
    GS_CHECK_ARG(grid_ok(R0, R1, R2), "bad resolution");
    const FcGrid g = make_grid(R0, R1, R2);
    const FcWs w = fc_ws(g, ws);
    GS_CHECK_ARG(ws_bytes >= w.bytes, "workspace too small");
    GS_CHECK_HIP(ensure_tables());
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fc_cube_count_kernel, dim3(w.nb_cube), dim3(FC_BLOCK), 0, s, g, sdf, w.blk_cube);
    hipLaunchKernelGGL(fc_scan_blocks_kernel<NCC>, dim3(1), dim3(FC_BLOCK), 0, s, w.nb_cube, w.blk_cube, w.hdr + H_CUBE_TOT);
    hipLaunchKernelGGL(fc_slot_count_kernel, dim3(w.nb_slot), dim3(FC_BLOCK), 0, s, g, sdf, w.blk_slot);
    hipLaunchKernelGGL(fc_scan_blocks_kernel<NCE>, dim3(1), dim3(FC_BLOCK), 0, s, w.nb_slot, w.blk_slot, w.hdr + H_SLOT_TOT);
    hipLaunchKernelGGL(fc_cube_assign_kernel, dim3(w.nb_cube), dim3(FC_BLOCK), 0, s, g, sdf, w.blk_cube, w.hdr, w.case8, w.vd_base,
                       w.k_base);
    hipLaunchKernelGGL(fc_slot_assign_kernel, dim3(w.nb_slot), dim3(FC_BLOCK), 0, s, g, sdf, w.blk_slot, w.hdr, w.quad_id);
    hipLaunchKernelGGL(fc_counts_kernel, dim3(1), dim3(1), 0, s, w.hdr, counts);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_flexicubes_fwd(int R0, int R1, int R2, const float* vertices, const float* sdf, const float* alpha,
                                 const float* beta, const float* gamma, float weight_scale, float sdf_eps, const void* ws,
                                 size_t ws_bytes, int64_t Q, int64_t num_quads, int64_t K, float* out_vertices,
                                 int64_t* out_faces, float* L_dev, void* stream)
{
    GS_CHECK_ARG(grid_ok(R0, R1, R2), "bad resolution");
    GS_CHECK_ARG(Q > 0 && num_quads >= 0 && K > 0 && Q + num_quads < ((int64_t)1 << 31), "bad counts");
    const FcGrid g = make_grid(R0, R1, R2);
    const FcWs w = fc_ws(g, (void*)ws);
    GS_CHECK_ARG(ws_bytes >= w.bytes, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const FcParams prm = {weight_scale, sdf_eps};
    hipLaunchKernelGGL(fc_dual_vertex_kernel, dim3(gs_cdiv(g.C, 256)), dim3(256), 0, s, g, prm, vertices, sdf, alpha, beta,
                       w.case8, w.vd_base, w.k_base, out_vertices, L_dev);
    if (num_quads > 0)
        hipLaunchKernelGGL(fc_quad_kernel, dim3(gs_cdiv(3 * (int64_t)g.Vg, 256)), dim3(256), 0, s, g, prm, sdf, gamma, w.case8,
                           w.vd_base, w.quad_id, (int)Q, out_vertices, out_faces);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_flexicubes_bwd(int R0, int R1, int R2, const float* vertices, const float* sdf, const float* alpha,
                                 const float* beta, const float* gamma, float weight_scale, float sdf_eps, const void* ws,
                                 size_t ws_bytes, int64_t Q, int64_t num_quads, int64_t K, const float* out_vertices,
                                 const float* v_out_vertices, const float* v_L_dev, float* g_vd_scratch, float* g_vertices,
                                 float* g_sdf, float* g_alpha, float* g_beta, float* g_gamma, void* stream)
{
    GS_CHECK_ARG(grid_ok(R0, R1, R2), "bad resolution");
    GS_CHECK_ARG(Q > 0 && num_quads >= 0 && K > 0, "bad counts");
    GS_CHECK_ARG((alpha != nullptr) == (g_alpha != nullptr) && (beta != nullptr) == (g_beta != nullptr) &&
                 (gamma != nullptr) == (g_gamma != nullptr), "gradient buffers must match the given weights");
    const FcGrid g = make_grid(R0, R1, R2);
    const FcWs w = fc_ws(g, (void*)ws);
    GS_CHECK_ARG(ws_bytes >= w.bytes, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const FcParams prm = {weight_scale, sdf_eps};
    GS_CHECK_HIP(hipMemcpyAsync(g_vd_scratch, v_out_vertices, sizeof(float) * 3 * (size_t)Q, hipMemcpyDeviceToDevice, s));
    GS_CHECK_HIP(gs_zero_async(g_vertices, sizeof(float) * 3 * (size_t)g.Vg, s));
    GS_CHECK_HIP(gs_zero_async(g_sdf, sizeof(float) * (size_t)g.Vg, s));
    if (g_alpha) GS_CHECK_HIP(gs_zero_async(g_alpha, sizeof(float) * 8 * (size_t)g.C, s));
    if (g_beta) GS_CHECK_HIP(gs_zero_async(g_beta, sizeof(float) * 12 * (size_t)g.C, s));
    if (g_gamma) GS_CHECK_HIP(gs_zero_async(g_gamma, sizeof(float) * (size_t)g.C, s));
    if (num_quads > 0)
        hipLaunchKernelGGL(fc_quad_bwd_kernel, dim3(gs_cdiv(3 * (int64_t)g.Vg, 256)), dim3(256), 0, s, g, prm, sdf, gamma, w.case8,
                           w.vd_base, w.quad_id, (int)Q, out_vertices, v_out_vertices, g_vd_scratch, g_gamma);
    hipLaunchKernelGGL(fc_dual_vertex_bwd_kernel, dim3(gs_cdiv(g.C, 256)), dim3(256), 0, s, g, prm, vertices, sdf, alpha, beta,
                       w.case8, w.vd_base, w.k_base, out_vertices, g_vd_scratch, v_L_dev, g_vertices, g_sdf, g_alpha, g_beta);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// entropy of the sign-changing grid edges; needs the totals gs_flexicubes_count left in `ws`
extern "C" int gs_flexicubes_entropy_fwd(int R0, int R1, int R2, const float* sdf, void* ws, size_t ws_bytes, float* out,
                                         void* stream)
{
    GS_CHECK_ARG(grid_ok(R0, R1, R2), "bad resolution");
    const FcGrid g = make_grid(R0, R1, R2);
    const FcWs w = fc_ws(g, ws);
    GS_CHECK_ARG(ws_bytes >= w.bytes, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fc_entropy_kernel, dim3(w.nb_ent), dim3(256), 0, s, g, sdf, w.partial);
    hipLaunchKernelGGL(fc_entropy_finalize_kernel, dim3(1), dim3(256), 0, s, w.nb_ent, w.partial, w.hdr, out);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_flexicubes_entropy_bwd(int R0, int R1, int R2, const float* sdf, const void* ws, size_t ws_bytes,
                                         const float* v_out, float* g_sdf, int accumulate, void* stream)
{
    GS_CHECK_ARG(grid_ok(R0, R1, R2), "bad resolution");
    const FcGrid g = make_grid(R0, R1, R2);
    const FcWs w = fc_ws(g, (void*)ws);
    GS_CHECK_ARG(ws_bytes >= w.bytes, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) GS_CHECK_HIP(gs_zero_async(g_sdf, sizeof(float) * (size_t)g.Vg, s));
    hipLaunchKernelGGL(fc_entropy_bwd_kernel, dim3(w.nb_ent), dim3(256), 0, s, g, sdf, w.hdr, v_out, g_sdf);
    GS_CHECK_LAUNCH();
    return GS_OK;
}
