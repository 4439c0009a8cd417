// gs_cube.h -- seam-aware bilinear cube-map addressing shared by the shading (S3) and prefilter (S5) kernels.
// Face order +x,-x,+y,-y,+z,-z with the (x,y) parameterisation of _cube_to_dir
// (rfstudio/graphics/_mesh/_texture.py:178-197) == cube_to_dir (rfstudio/graphics/_mesh/_splitsum/c_src/cubemap.cu:32-46).
// Texel semantics (edge wrap by re-projection, corner = mean of the other three) are documented in
// oracle/gs_oracle_shade.c.
#pragma once
#include "gs_common.h"

struct FaceMap { int a, b, c; float sx, sy; };
// The per-face axis map  { a, b, c, sx, sy }:
//   +x {2,1,0,-1,-1}  -x {2,1,0,+1,-1}  +y {0,2,1,+1,+1}  -y {0,2,1,+1,-1}  +z {0,1,2,+1,-1}  -z {0,1,2,-1,-1}
// computed, not looked up: the face index differs per lane, so a __constant__ table is a VECTOR memory load -- and its
// s_waitcnt also waits for every texel tap issued before it (loads return in order), which chained the taps of
// successive cube lookups one after the other.
__device__ __forceinline__ FaceMap face_map(int s)
{
    const int axis = s >> 1;
    FaceMap m;
    m.a = axis == 0 ? 2 : 0;
    m.b = axis == 1 ? 2 : 1;
    m.c = axis;
    m.sx = (s == 0 || s == 5) ? -1.0f : 1.0f;
    m.sy = s == 2 ? 1.0f : -1.0f;
    return m;
}

__device__ __forceinline__ int select_face(const float* d)
{
    const float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]);
    int f; float c;
    if (az > fmaxf(ax, ay)) { f = 4; c = d[2]; }
    else if (ay > ax)       { f = 2; c = d[1]; }
    else                    { f = 0; c = d[0]; }
    return f + (c < 0.0f ? 1 : 0);
}

__device__ __forceinline__ void face_point(int s, float x, float y, float* p)
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

__device__ __forceinline__ float comp3(const float* d, int i) { return i == 0 ? d[0] : (i == 1 ? d[1] : d[2]); }

// Texel (ix, iy) of face s where one coordinate lies outside [0, R): the reference rule (oracle/gs_oracle_shade.c) pushes the texel
// CENTRE through face_point / select_face / the new face's map and rounds.  Kept as the definition (self-test only).
__device__ __attribute__((noinline)) int resolve_texel_reproject(int s, int ix, int iy, int R)
{
    const float xn = 2.0f * (((float)ix + 0.5f) / (float)R) - 1.0f;
    const float yn = 2.0f * (((float)iy + 0.5f) / (float)R) - 1.0f;
    float p[3];
    face_point(s, xn, yn, p);
    const int s2 = select_face(p);
    const FaceMap m = face_map(s2);
    const float inv = 1.0f / fabsf(comp3(p, m.c));
    const float x2 = m.sx * comp3(p, m.a) * inv, y2 = m.sy * comp3(p, m.b) * inv;
    const float tx = (x2 + 1.0f) * 0.5f * (float)R - 0.5f, ty = (y2 + 1.0f) * 0.5f * (float)R - 0.5f;
    int jx = (int)floorf(tx + 0.5f), jy = (int)floorf(ty + 0.5f);
    jx = min(max(jx, 0), R - 1); jy = min(max(jy, 0), R - 1);
    return (s2 * R + jy) * R + jx;
}

// The re-projection always lands on the texel ADJACENT across the edge: along the edge the centre moves by less than
// R / (2 (R + 1)) < 1/2 texel, across it the result is the first or last row / column of the neighbour.  So it is an integer map
// (face, edge) -> (face', x-kind, y-kind) with the kinds {0, R - 1, t, R - 1 - t} of the in-range coordinate t, independent of R:
// 24 entries of 7 bits, one 32-bit constant per face (edge e = {ix < 0, ix >= R, iy < 0, iy >= R} at bits 7e: face' | x-kind << 3 |
// y-kind << 5).  Derived from the float rule and compared with it for EVERY (face, edge, t) at R = 2..64 and 22 sizes up to 5000
// (on the device: gs_selftest_cube_edges, tests/test_gpu_shading.py); at R = 8192 the float rule itself runs out of precision,
// and every entry that takes a cube map rejects faces beyond 4096^2 (env_to_dev).  Round 4: the float rule (three IEEE divisions,
// ~100 instructions per corner) ran in nearly every wave of the shading kernels -- one lane of 64 on a face edge is enough.
#define GS_CUBE_EDGE_TABLE_MAX_R 4096
__device__ __forceinline__ int resolve_texel(int s, int ix, int iy, int R)
{
    const bool ox = (ix < 0 || ix >= R), oy = (iy < 0 || iy >= R);
    if (!ox && !oy) return (s * R + iy) * R + ix;
    if (ox && oy) return -1;
    const int e = ox ? (ix < 0 ? 0 : 1) : (iy < 0 ? 2 : 3);
    const int t = ox ? iy : ix;
    const unsigned w = s == 0 ? 0x97aa2ccu : s == 1 ? 0xc70a24du : s == 2 ? 0x2874c11u : s == 3 ? 0x7ad1839u : s == 4 ? 0x26ca049u : 0x766a0c8u;
    const unsigned c = (w >> (7 * e)) & 127u;
    const int s2 = (int)(c & 7u), kx = (int)((c >> 3) & 3u), ky = (int)((c >> 5) & 3u);
    const int jx = kx == 0 ? 0 : (kx == 1 ? R - 1 : (kx == 2 ? t : R - 1 - t));
    const int jy = ky == 0 ? 0 : (ky == 1 ? R - 1 : (ky == 2 ? t : R - 1 - t));
    return (s2 * R + jy) * R + jx;
}

struct CubeFp {
    bool valid;
    int idx[4];
    float w[4];
    float fx, fy;
    int face;
    float inv_c, xn, yn;
};

__device__ void cube_footprint(const float* d, int R, CubeFp& fp)
{
    const int s = select_face(d);
    const FaceMap m = face_map(s);
    const float ac = fabsf(comp3(d, m.c));
    fp.valid = (ac > 0.0f) && isfinite(ac);
    fp.face = s;
    if (!fp.valid) { fp.idx[0] = fp.idx[1] = fp.idx[2] = fp.idx[3] = 0; return; }
    const float inv = 1.0f / ac;
    const float xn = m.sx * comp3(d, m.a) * inv, yn = m.sy * comp3(d, m.b) * inv;
    fp.inv_c = inv; fp.xn = xn; fp.yn = yn;
    const float tx = (xn + 1.0f) * 0.5f * (float)R - 0.5f, ty = (yn + 1.0f) * 0.5f * (float)R - 0.5f;
    const int ix0 = (int)floorf(tx), iy0 = (int)floorf(ty);
    const float fx = tx - (float)ix0, fy = ty - (float)iy0;
    fp.fx = fx; fp.fy = fy;
    if (ix0 >= 0 && iy0 >= 0 && ix0 + 1 < R && iy0 + 1 < R) {       // interior fast path
        const int b = (s * R + iy0) * R + ix0;
        fp.idx[0] = b; fp.idx[1] = b + 1; fp.idx[2] = b + R; fp.idx[3] = b + R + 1;
    } else {
        fp.idx[0] = resolve_texel(s, ix0, iy0, R);
        fp.idx[1] = resolve_texel(s, ix0 + 1, iy0, R);
        fp.idx[2] = resolve_texel(s, ix0, iy0 + 1, R);
        fp.idx[3] = resolve_texel(s, ix0 + 1, iy0 + 1, R);
    }
    float w[4] = { (1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy };
    int miss = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (fp.idx[i] < 0) miss = i;
#pragma unroll
    for (int i = 0; i < 4; ++i) fp.w[i] = w[i];
    if (miss >= 0) {
        float wm = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i == miss) wm = w[i] / 3.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) fp.w[i] = (i == miss) ? 0.0f : w[i] + wm;
    }
}

// bilinear cube fetch (3 channels); WITH_GRAD also returns d out / d dir (dd[c*3+k])
template <bool WITH_GRAD>
__device__ void cube_fetch(const float* __restrict__ tex, int R, const float* d, float* out, float* dd, CubeFp& fp)
{
    cube_footprint(d, R, fp);
    // BRANCH-FREE loads: a load under a condition makes the compiler drain vmcnt at the join, which turned the 8 taps of a
    // trilinear sample into 8 sequential L2 round trips.  Invalid footprints and the missing corner texel read element 0
    // and are zeroed by selects afterwards.
    const bool valid = fp.valid;
    bool ok[4]; int safe[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ok[i] = valid && fp.idx[i] >= 0; safe[i] = ok[i] ? fp.idx[i] : 0; }
    float t[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
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
            const float s = (t[0][c] + t[1][c] + t[2][c] + t[3][c]) / 3.0f;   // the missing one holds 0
#pragma unroll
            for (int i = 0; i < 4; ++i) if (fp.idx[i] < 0) t[i][c] = s;
        }
    }
    if (!valid) {
        out[0] = out[1] = out[2] = 0.0f;
        if (WITH_GRAD) { for (int i = 0; i < 9; ++i) dd[i] = 0.0f; }
        return;
    }
    const FaceMap m = face_map(fp.face);
    const float sgn_c = comp3(d, m.c) < 0.0f ? -1.0f : 1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = t[0][c] + fp.fx * (t[1][c] - t[0][c]);
        const float bot = t[2][c] + fp.fx * (t[3][c] - t[2][c]);
        out[c] = top + fp.fy * (bot - top);
        if (WITH_GRAD) {
            const float dtx = (t[1][c] - t[0][c]) + fp.fy * ((t[3][c] - t[2][c]) - (t[1][c] - t[0][c]));
            const float dty = bot - top;
            const float gx = dtx * 0.5f * (float)R, gy = dty * 0.5f * (float)R;
            float g[3] = { 0.0f, 0.0f, 0.0f };
            const float va = gx * m.sx * fp.inv_c, vb = gy * m.sy * fp.inv_c;
            const float vc = -(gx * fp.xn + gy * fp.yn) * fp.inv_c * sgn_c;
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] = (k == m.a ? va : 0.0f) + (k == m.b ? vb : 0.0f) + (k == m.c ? vc : 0.0f);
            dd[c * 3 + 0] = g[0]; dd[c * 3 + 1] = g[1]; dd[c * 3 + 2] = g[2];
        }
    }
}

// ---- wave-aggregated texel-gradient scatter ---------------------------------------------------------------
// Neighbouring Gaussians of a surface reflect into the same few texels of the coarse pyramid levels, so the
// 64 lanes of a wave would send many fp32 atomics at identical addresses (the L2 serialises them: 1.26 ms of
// the 1.3 ms shading backward at 2 M Gaussians).  Lanes that target the SAME address are summed in registers
// (leader loop: broadcast the first pending address, ballot the matches, DPP-reduce, one lane commits);
// incoherent lanes fall back to plain atomics after GS_AGG_ROUNDS leaders.
#ifndef GS_AGG_ROUNDS
#define GS_AGG_ROUNDS 0    // round 1 (permute commit): 0 / 1 / 2 / 6 rounds = 0.807 / 0.802 / 0.833 / 0.846 ms (kernel + glue); round 2 (LDS-staged commit, kernel alone): 0 / 1 rounds = 364 / 381 us
#endif
template <bool XCD_LOCAL>
__device__ __forceinline__ void gs_add_scoped(float* p, float v)
{
    if (XCD_LOCAL) gs_atomic_add_xcd(p, v); else gs_atomic_add(p, v);
}

// Row pairs (the per-texel transposed commit of round 1 is gone): the two taps of one bilinear ROW are neighbours in memory (texel x and x+1 of a face row = six
// contiguous floats), so their six atomics are issued by SIX adjacent lanes of one instruction and merge into ONE
// memory-side request (two when the 24 bytes straddle a cache line or, on a face edge, the second texel lives
// elsewhere).  Halves the request count of the shading backward, which sits at the fabric's atomic request rate.
// LDS-staged form of wave_commit6 (below): every lane writes its two pointers and six values into a wave-private
// [10 words][64 lanes] tile, then reads the (pointer, value) pair it has to issue -- 8 writes + 12 reads per call instead of
// 60 ds_bpermute, which made the LDS pipe of the CU (8.3 M LDS instructions per launch, 7.4 M of them permutes) a bottleneck
// of the shading backward.  `stage` = 640 floats of LDS owned by this wave; all 64 lanes call together.
template <bool XCD_LOCAL>
__device__ __forceinline__ void wave_commit6_lds(float* stage, float* pa, float* pb, const float (&v)[6])
{
    const int lane = (int)(threadIdx.x & 63);
    unsigned long long* sp = reinterpret_cast<unsigned long long*>(stage);       // [2][64] pointers, then [6][64] values
    float* sv = stage + 256;
    __builtin_amdgcn_wave_barrier();
    sp[lane] = (unsigned long long)pa; sp[64 + lane] = (unsigned long long)pb;
#pragma unroll
    for (int c = 0; c < 6; ++c) sv[c * 64 + lane] = v[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int q = 64 * k + lane;
        const int src = q / 6, ch = q - 6 * src;
        const bool second = ch >= 3;
        float* base = (float*)sp[(second ? 64 : 0) + src];
        const float val = sv[ch * 64 + src];
#ifndef GS_EXPERIMENT_NO_GLOBAL_TEXEL_ATOMICS
        if (base != nullptr) gs_add_scoped<XCD_LOCAL>(base + (second ? ch - 3 : ch), val);
#else
        if (base != nullptr && val == 123456.0f) base[0] = val;      /* timing experiment only */
#endif
    }
}

template <bool XCD_LOCAL>
__device__ __forceinline__ void wave_commit6(float* pa, float* pb, const float (&v)[6])
{
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long ka = (unsigned long long)pa, kb = (unsigned long long)pb;
    const int alo = (int)(unsigned)ka, ahi = (int)(unsigned)(ka >> 32), blo = (int)(unsigned)kb, bhi = (int)(unsigned)(kb >> 32);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int q = 64 * k + lane;
        const int src = q / 6, ch = q - 6 * src;
        const int sa = src << 2;
        // ds_bpermute: this lane READS the operand of lane `src`; fetch both pointers and all six values, pick by ch
        const unsigned a_lo = (unsigned)__builtin_amdgcn_ds_bpermute(sa, alo), a_hi = (unsigned)__builtin_amdgcn_ds_bpermute(sa, ahi);
        const unsigned b_lo = (unsigned)__builtin_amdgcn_ds_bpermute(sa, blo), b_hi = (unsigned)__builtin_amdgcn_ds_bpermute(sa, bhi);
        float x[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) x[c] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sa, __builtin_bit_cast(int, v[c])));
        const bool second = ch >= 3;
        float* base = (float*)(((unsigned long long)(second ? b_hi : a_hi) << 32) | (second ? b_lo : a_lo));
        const int c3 = second ? ch - 3 : ch;
        float val = x[0];
#pragma unroll
        for (int c = 1; c < 6; ++c) val = (ch == c) ? x[c] : val;
#ifndef GS_EXPERIMENT_NO_GLOBAL_TEXEL_ATOMICS
        if (base != nullptr) gs_add_scoped<XCD_LOCAL>(base + c3, val);
#else
        if (base != nullptr && val == 123456.0f) base[c3] = val;      /* timing experiment only */
#endif
    }
}

// aggregate lanes that hit the same row pair (same first AND second texel), then commit transposed
template <bool XCD_LOCAL>
__device__ __forceinline__ void wave_agg_add6(float* pa, float* pb, const float (&v)[6], float* stage = nullptr)
{
    const int lane = (int)(threadIdx.x & 63);
    unsigned long long ka = (unsigned long long)pa;
    const unsigned long long kb = (unsigned long long)pb;
    unsigned long long remaining = __ballot(pa != nullptr || pb != nullptr);
    int singles = 0;
    float* enda = nullptr; float* endb = nullptr;
    float pv[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
    bool mine = (pa != nullptr || pb != nullptr);
    for (int round = 0; round < GS_AGG_ROUNDS && remaining != 0ull && singles < 2; ++round) {
        const int leader = __builtin_ctzll(remaining);
        const unsigned long long la = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(ka >> 32), leader) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)(unsigned)ka, leader);
        const unsigned long long lb = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(kb >> 32), leader) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)(unsigned)kb, leader);
        const bool same = mine && ka == la && kb == lb;
        const unsigned long long m = __ballot(same);
        float s[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) s[c] = gs_wave_sum(same ? v[c] : 0.0f);
        if (lane == leader) {
            enda = pa; endb = pb;
#pragma unroll
            for (int c = 0; c < 6; ++c) pv[c] = s[c];
        }
        if (same) mine = false;
        remaining &= ~m;
        singles = (__popcll(m) == 1) ? singles + 1 : 0;
    }
    if (mine) {                                        // not aggregated: commit the lane's own contribution
        enda = pa; endb = pb;
#pragma unroll
        for (int c = 0; c < 6; ++c) pv[c] = v[c];
    }
    if (stage) wave_commit6_lds<XCD_LOCAL>(stage, enda, endb, pv);
    else wave_commit6<XCD_LOCAL>(enda, endb, pv);
}

// all 64 lanes of the wave must call this together (lanes without work pass valid == false)
template <bool XCD_LOCAL>
__device__ __forceinline__ void cube_scatter_wave(float* grad_tex, const CubeFp& fp, const float* g, float scale, bool valid,
                                                  float* stage = nullptr)
{
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        float* p[2]; float v[6];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * row + j;
            const bool on = valid && fp.valid && fp.idx[i] >= 0 && grad_tex != nullptr;
            const float w = on ? scale * fp.w[i] : 0.0f;
            p[j] = on ? grad_tex + (size_t)fp.idx[i] * 3 : nullptr;
            v[3 * j] = g[0] * w; v[3 * j + 1] = g[1] * w; v[3 * j + 2] = g[2] * w;
        }
        wave_agg_add6<XCD_LOCAL>(p[0], p[1], v, stage);
    }
}

// ---- tagged variant (round 4): the scope of the atomic is a per-LANE property --------------------------------------------------
// The fused tail keeps XCD-private copies only of the MID-SIZED pyramid levels (64^2, 128^2: 60 % of the row-pair requests of the
// bench scene, 1.5 MB per copy -- zeroed and folded once per STEP), whose atomics then resolve in the XCD's own L2 instead of at
// the memory side of the fabric; the large levels (256^2, 512^2) stay device-scope.  Which of the two a lane needs depends on ITS
// mip level, so the pointer carries the choice in bit 0 (texel addresses are 4-byte aligned).
__device__ __forceinline__ void wave_commit6_lds_tagged(float* stage, float* pa, float* pb, const float (&v)[6])
{
    // COMPACTED: only the lanes that have a row pair enter the staging (at their rank), and the transposed commit issues
    // ceil(6 n / 64) atomic instructions instead of six -- in the pair kernels 60 % of the level fetches go to the LDS copies, so a
    // commit carries ~25 row pairs, and an atomic instruction costs its issue slot whatever its exec mask holds.
    const int lane = (int)(threadIdx.x & 63);
    unsigned long long* sp = reinterpret_cast<unsigned long long*>(stage);       // [2][64] pointers, then [6][64] values
    float* sv = stage + 256;
    const bool act = pa != nullptr || pb != nullptr;
    const unsigned long long am = __ballot(act);
    if (am == 0ull) return;
    const int n6 = 6 * __popcll(am);
    const int rank = __popcll(am & ((1ull << lane) - 1ull));
    __builtin_amdgcn_wave_barrier();
    if (act) {
        sp[rank] = (unsigned long long)pa; sp[64 + rank] = (unsigned long long)pb;
#pragma unroll
        for (int c = 0; c < 6; ++c) sv[c * 64 + rank] = v[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int k = 0; 64 * k < n6; ++k) {
        const int q = 64 * k + lane;
        if (q < n6) {
            const int src = q / 6, ch = q - 6 * src;
            const bool second = ch >= 3;
            const unsigned long long tagged = sp[(second ? 64 : 0) + src];
            float* base = (float*)(tagged & ~1ull);
            const float val = sv[ch * 64 + src];
#ifndef GS_EXPERIMENT_NO_GLOBAL_TEXEL_ATOMICS
            if (base != nullptr) {
                float* dst = base + (second ? ch - 3 : ch);
                if (tagged & 1ull) gs_atomic_add_xcd(dst, val); else gs_atomic_add(dst, val);
            }
#else
            if (base != nullptr && val == 123456.0f) base[0] = val;      /* timing experiment only */
#endif
        }
    }
}

// all 64 lanes of the wave must call this together; `local`: grad_tex is this XCD's private copy
__device__ __forceinline__ void cube_scatter_wave_tagged(float* grad_tex, bool local, const CubeFp& fp, const float* g, float scale, bool valid,
                                                         float* stage)
{
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        float* p[2]; float v[6];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * row + j;
            const bool on = valid && fp.valid && fp.idx[i] >= 0 && grad_tex != nullptr;
            const float w = on ? scale * fp.w[i] : 0.0f;
            p[j] = on ? (float*)((unsigned long long)(grad_tex + (size_t)fp.idx[i] * 3) | (local ? 1ull : 0ull)) : nullptr;
            v[3 * j] = g[0] * w; v[3 * j + 1] = g[1] * w; v[3 * j + 2] = g[2] * w;
        }
        wave_commit6_lds_tagged(stage, p[0], p[1], v);
    }
}
