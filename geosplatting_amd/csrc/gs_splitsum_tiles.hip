// gs_splitsum_tiles.hip -- S5: the specular split-sum prefilter as a TILED sparse operator (levels with R >= 64).
//
// What it computes is SpecularCubemapFwd/BwdKernel (rfstudio/graphics/_mesh/_splitsum/c_src/cubemap.cu:246-350):
//     out[o] = sum_i w(o,i) cubemap[i] / sum_i w(o,i),      w(o,i) = g(o,i) * pixel_area(i) / 4   for the texels i of o's lobe.
// w depends on (R, roughness, cutoff) only while the cubemap changes every training step, so the weights are kept -- but not as
// one list per texel (round 2: 13 GB, 11.8 GB streamed per step, one wave per texel gathering its sources from L1).  Here:
//
//  * lane = output texel.  A wave owns an 8x8 block of output texels, a workgroup a 16x16 tile (four blocks).  For a source face
//    every lane walks ITS OWN lobe box from the box's corner ("anchor"), all lanes in lock step: step (dx, dy) pairs lane l with
//    source anchor_l + (dx, dy).  Neighbouring lobes are shifted copies of each other, so a row of 64 weights (one per lane) is
//    80-90 % dense -- the per-texel 8x8 patches of round 2 were 35-60 % dense -- and nothing is reduced across lanes.
//  * the sources of a tile's lobes on one face are a rectangle, staged ONCE per tile and face in LDS (premultiplied by
//    pixel_area / 4) and read with one `ds_read_b128` per lane and row: no gathers from L1, which bounded the round-2 kernel.
//  * g (gs_splitsum_math.h) is invariant under the cube's three reflections (sign flips of a world axis: exact in binary
//    floating point), so ONE table row serves the eight mirror images of a block: the table covers one octant of the cube
//    (1/8 of the rows: 1.1 GB instead of 13 GB for a 512^2 pyramid), eight workgroups with consecutive ids on the same XCD
//    stream the same rows (one HBM read, seven L2 hits) and differ only in the reflection applied to the texel indices they
//    stage and write.  pixel_area is NOT reflection symmetric (cubemap.cu:17-30 measures |x - R/2| from the texel's lower
//    edge on one side and its upper edge on the other): it is applied to the staged sources, per mirror.
//    gs_specular_tiles_check verifies on the device what the construction relies on (direction table and lobe boxes are exact
//    mirror images); a level that fails runs the same kernels on tables built for all texels (n_mirrors = 1).
//  * the backward is the forward's transpose as a GATHER (lane = source texel, partners = the outputs whose box contains it:
//    tested pair by pair against the outputs' own boxes, so it is the exact adjoint although the reference's tile-culled boxes
//    are not exactly symmetric): no atomics, bit-reproducible.
//
// Cost model: per 64-weight row one 256-byte weight load, one ds_read_b128 (4 LDS cycles), ~5 VALU.
#include "gs_common.h"

#pragma clang fp contract(off)
#include "gs_splitsum_math.h"

typedef float gs_f2 __attribute__((ext_vector_type(2)));
typedef float gs_f4 __attribute__((ext_vector_type(4)));
#ifndef GS_TILE_DEPTH
#define GS_TILE_DEPTH 4             // register sets of 8 rows in flight per wave (8: measured slower)
#endif
#define GS_TILE_ROW_PAD 8          // rows of a (block, face) list are padded to a multiple of this (zero weights): no tail branches

// texel index of the image of (s, x, y) under the reflection m (bit k of m = flip world axis k).  Face (x, y) -> direction:
// gs_cube.h face_point.  +-x faces: y <-> axis 1, x <-> axis 2; +-y: x <-> axis 0, y <-> axis 2; +-z: x <-> axis 0, y <-> axis 1;
// flipping the face's own axis swaps s with s ^ 1 and reflects the coordinate whose sign differs between the two faces.
__device__ __forceinline__ int mirror_texel(int s, int x, int y, int m, int R)
{
    const int pair = s >> 1;
    const int ax = pair == 0 ? 2 : 0;
    const int ay = pair == 1 ? 2 : 1;
    int fx = (m >> ax) & 1, fy = (m >> ay) & 1;
    if ((m >> pair) & 1) {
        s ^= 1;
        if (pair == 1) fy ^= 1; else fx ^= 1;
    }
    x = fx ? R - 1 - x : x;
    y = fy ? R - 1 - y : y;
    return (s * R + y) * R + x;
}

__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

struct LaneBox { int xmin, xmax, ymin, ymax; bool any; };

// lobe box of texel t on face `face` (gs_specular_bounds layout [6R^2][6][4] = xmin, xmax, ymin, ymax as floats), grown by E
__device__ __forceinline__ LaneBox lane_box(const float* __restrict__ bounds, int t, int face, int E, int R)
{
    const float4 b = *reinterpret_cast<const float4*>(bounds + (size_t)t * 24 + face * 4);
    LaneBox r;
    r.xmin = (int)b.x; r.xmax = (int)b.y; r.ymin = (int)b.z; r.ymax = (int)b.w;
    r.any = r.xmin <= r.xmax && r.ymin <= r.ymax;
    if (E >= R) { r.xmin = r.ymin = 0; r.xmax = r.ymax = R - 1; r.any = true; return r; }     // every texel of the face is a candidate
    if (r.any && E > 0) {
        r.xmin = max(r.xmin - E, 0); r.xmax = min(r.xmax + E, R - 1);
        r.ymin = max(r.ymin - E, 0); r.ymax = min(r.ymax + E, R - 1);
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Table construction.  One workgroup per (tile, source face), wave b = block b of the tile (nb = bw x bh blocks of 8x8 texels).  Rows are enumerated dy-major over
// the largest box of the block; a row is kept if any lane has a partner in it.
//   MODE 0: count the kept rows (+ the extent of the block's source region, + the number of pairs)
//   MODE 1: write descriptors ((dy * pitch + dx) * 16: the row's byte offset in the staged rectangle) and weights g (0 where
//           the lane has no partner)
// FWD rows: lane = output o (VNR), partner = source i (L) inside o's box with L.VNR >= cutoff (exactly the forward loop).
// BWD rows: lane = source i (L), partners = outputs o (VNR) inside i's box grown by E with L.VNR >= cutoff AND i inside o's box.
template <bool BWD, int MODE>
__global__ void __launch_bounds__(1024)
tile_build_kernel(int R, float roughness, float cutoff, int E, int bw, int nb, const float* __restrict__ bounds,
                  const float4* __restrict__ table, const int4* __restrict__ tiles, int32_t* __restrict__ cnt /*[nt][6][nb]*/,
                  int4* __restrict__ ext /*[nt][6][nb]*/, unsigned long long* __restrict__ pairs,
                  const int64_t* __restrict__ row_begin /*[nt][6][nb]*/, const int4* __restrict__ seg /*[nt][6]*/,
                  int32_t* __restrict__ desc, float* __restrict__ weights)
{
    const int lane = threadIdx.x & 63;
    const int b = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tile = blockIdx.x / 6, face = blockIdx.x % 6;
    const int4 tl = tiles[tile];
    const int own_s = tl.x, own_x = tl.y + 8 * (b % bw) + (lane & 7), own_y = tl.z + 8 * (b / bw) + (lane >> 3);
    const int t = (own_s * R + own_y) * R + own_x;
    const float4 own4 = table[t];
    const float own[3] = { own4.x, own4.y, own4.z };
    const float alpha = roughness * roughness;
    const float alphaSqr = alpha * alpha;
    const LaneBox bx = lane_box(bounds, t, face, E, R);
    const int W = wave_max_i(bx.any ? bx.xmax - bx.xmin + 1 : 0);
    const int H = wave_max_i(bx.any ? bx.ymax - bx.ymin + 1 : 0);
    const int slot = (tile * 6 + face) * nb + b;
    int kept = 0;
    unsigned long long npairs = 0;
    int64_t row0 = 0;
    int pitch = 0;
    if (MODE == 1) { row0 = row_begin[slot]; pitch = seg[tile * 6 + face].w; }
    for (int dy = 0; dy < H; ++dy)
        for (int dx = 0; dx < W; ++dx) {
            bool in = bx.any && dx <= bx.xmax - bx.xmin && dy <= bx.ymax - bx.ymin;
            float g = 0.0f;
            if (in) {
                const int px = bx.xmin + dx, py = bx.ymin + dy;
                const size_t ti = ((size_t)face * R + py) * R + px;
                const float4 o4 = table[ti];
                const float other[3] = { o4.x, o4.y, o4.z };
                const float* L = BWD ? own : other;
                const float* VNR = BWD ? other : own;
                const float ldv = dot3(L, VNR);
                in = ldv >= cutoff;
                if (BWD && in) {
                    const float4 ob = *reinterpret_cast<const float4*>(bounds + ti * 24 + own_s * 4);
                    in = own_x >= (int)ob.x && own_x <= (int)ob.y && own_y >= (int)ob.z && own_y <= (int)ob.w;
                }
                if (MODE == 1 && in) g = specular_pair_g(L, VNR, ldv, alphaSqr);
            }
            const unsigned long long mask = __ballot(in);
            if (mask != 0ull) {
                if (MODE == 1) {
                    const int64_t row = row0 + kept;
                    weights[(row >> 1) * 128 + lane * 2 + (row & 1)] = g;     // rows in pairs: one 8-byte load per lane and two rows
                    if (lane == 0) desc[row] = (dy * pitch + dx) * 16;     // byte offset from the lane's anchor in the staged rectangle
                }
                ++kept;
                npairs += (unsigned long long)__popcll(mask);
            }
        }
    if (MODE == 0) {
        const int minx = wave_min_i(bx.any ? bx.xmin : 0x7fffffff), maxa = wave_max_i(bx.any ? bx.xmin : -1);
        const int miny = wave_min_i(bx.any ? bx.ymin : 0x7fffffff), mayb = wave_max_i(bx.any ? bx.ymin : -1);
        if (lane == 0) {
            cnt[slot] = kept;
            // region a row address can reach: [min anchor, max anchor + (W - 1, H - 1)]
            ext[slot] = kept > 0 ? make_int4(minx, maxa + W - 1, miny, mayb + H - 1) : make_int4(0x7fffffff, -0x7fffffff, 0x7fffffff, -0x7fffffff);
            if (npairs) atomicAdd(pairs, npairs);
        }
    }
}

static bool tile_geometry_ok(int R, int bw, int nb)
{
    return R >= 16 && (R % 16) == 0 && bw >= 1 && nb >= 1 && nb <= 16 && (nb % bw) == 0;
}

extern "C" int gs_specular_tiles_count(int R, float roughness, float costheta_cutoff, int backward, int margin, int bw, int nb,
                                       const float* bounds, const float* dir_table, const int32_t* tiles, int n_tiles,
                                       int32_t* row_counts, int32_t* extents, uint64_t* pairs, void* stream)
{
    GS_CHECK_ARG(tile_geometry_ok(R, bw, nb) && n_tiles >= 0 && margin >= 0, "R must be a multiple of 16, 1 <= nb <= 16 blocks in rows of bw");
    GS_CHECK_ARG(bounds && dir_table && tiles && row_counts && extents && pairs, "null argument");
    if (n_tiles == 0) return GS_OK;
    const dim3 grid(n_tiles * 6), block(64 * nb);
    if (backward)
        hipLaunchKernelGGL((tile_build_kernel<true, 0>), grid, block, 0, (hipStream_t)stream, R, roughness, costheta_cutoff, margin, bw, nb,
                           bounds, (const float4*)dir_table, (const int4*)tiles, row_counts, (int4*)extents, (unsigned long long*)pairs,
                           nullptr, nullptr, nullptr, nullptr);
    else
        hipLaunchKernelGGL((tile_build_kernel<false, 0>), grid, block, 0, (hipStream_t)stream, R, roughness, costheta_cutoff, 0, bw, nb,
                           bounds, (const float4*)dir_table, (const int4*)tiles, row_counts, (int4*)extents, (unsigned long long*)pairs,
                           nullptr, nullptr, nullptr, nullptr);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_specular_tiles_fill(int R, float roughness, float costheta_cutoff, int backward, int margin, int bw, int nb,
                                      const float* bounds, const float* dir_table, const int32_t* tiles, int n_tiles,
                                      const int64_t* row_begin, const int32_t* segments, int32_t* desc, float* weights, void* stream)
{
    GS_CHECK_ARG(tile_geometry_ok(R, bw, nb) && n_tiles >= 0 && margin >= 0, "R must be a multiple of 16, 1 <= nb <= 16 blocks in rows of bw");
    GS_CHECK_ARG(bounds && dir_table && tiles && row_begin && segments && desc && weights, "null argument");
    if (n_tiles == 0) return GS_OK;
    const dim3 grid(n_tiles * 6), block(64 * nb);
    if (backward)
        hipLaunchKernelGGL((tile_build_kernel<true, 1>), grid, block, 0, (hipStream_t)stream, R, roughness, costheta_cutoff, margin, bw, nb,
                           bounds, (const float4*)dir_table, (const int4*)tiles, nullptr, nullptr, nullptr, row_begin, (const int4*)segments,
                           desc, weights);
    else
        hipLaunchKernelGGL((tile_build_kernel<false, 1>), grid, block, 0, (hipStream_t)stream, R, roughness, costheta_cutoff, 0, bw, nb,
                           bounds, (const float4*)dir_table, (const int4*)tiles, nullptr, nullptr, nullptr, row_begin, (const int4*)segments,
                           desc, weights);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// What the shared-octant tables rely on, checked texel by texel on the device:
//   out[0] = texels whose mirrored direction-table entry is not the exact sign-flipped copy (all 7 reflections),
//   out[1] = (texel, face, reflection) triples whose lobe box is not the reflected box of the mirrored texel.
__global__ void __launch_bounds__(256)
tile_symmetry_check_kernel(int R, const float* __restrict__ bounds, const float4* __restrict__ table, unsigned long long* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 6 * R * R) return;
    const int x = t % R, y = (t / R) % R, s = t / (R * R);
    const float4 d = table[t];
    unsigned bad_dir = 0, bad_box = 0;
    for (int m = 1; m < 8; ++m) {
        const float4 q = table[mirror_texel(s, x, y, m, R)];
        const float ex = (m & 1) ? -d.x : d.x, ey = (m & 2) ? -d.y : d.y, ez = (m & 4) ? -d.z : d.z;
        bad_dir += (__float_as_uint(q.x) != __float_as_uint(ex)) || (__float_as_uint(q.y) != __float_as_uint(ey)) ||
                   (__float_as_uint(q.z) != __float_as_uint(ez));
        const int tm = mirror_texel(s, x, y, m, R);
        for (int f = 0; f < 6; ++f) {
            const LaneBox a = lane_box(bounds, t, f, 0, R);
            // the image of face f's box: corners (xmin, ymin) and (xmax, ymax) map to two opposite corners of the mirrored box
            const int c0 = mirror_texel(f, a.any ? a.xmin : 0, a.any ? a.ymin : 0, m, R);
            const int c1 = mirror_texel(f, a.any ? a.xmax : 0, a.any ? a.ymax : 0, m, R);
            const int fm = c0 / (R * R);
            const LaneBox bm = lane_box(bounds, tm, fm, 0, R);
            if (a.any != bm.any) { ++bad_box; continue; }
            if (!a.any) continue;
            const int x0 = c0 % R, y0 = (c0 / R) % R, x1 = c1 % R, y1 = (c1 / R) % R;
            bad_box += (min(x0, x1) != bm.xmin) || (max(x0, x1) != bm.xmax) || (min(y0, y1) != bm.ymin) || (max(y0, y1) != bm.ymax);
        }
    }
    if (bad_dir) atomicAdd(out, (unsigned long long)bad_dir);
    if (bad_box) atomicAdd(out + 1, (unsigned long long)bad_box);
}

extern "C" int gs_specular_tiles_check(int R, const float* bounds, const float* dir_table, uint64_t* out2, void* stream)
{
    GS_CHECK_ARG(R >= 1 && bounds && dir_table && out2, "bad arguments");
    GS_CHECK_HIP(gs_zero_async(out2, 2 * sizeof(uint64_t), (hipStream_t)stream));
    hipLaunchKernelGGL(tile_symmetry_check_kernel, dim3(gs_cdiv(6 * R * R, 256)), dim3(256), 0, (hipStream_t)stream, R, bounds,
                       (const float4*)dir_table, (unsigned long long*)out2);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Application.  One workgroup of 16 waves per (tile, reflection).  A tile is nb = bw x bh blocks; K = 16 / nb waves share a block and
// take its 8-row chunks round-robin (chunk c -> wave part c % K), their partial sums meet in LDS in a fixed order.  The geometry is
// the host's choice per level (splitsum.py): large tiles where the lobes are small against the face (the staged rectangle grows by
// the lobe diameter only once), one block split over 16 waves where a lobe covers most of a face and the level has few texels.
// With 8 reflections the workgroups of a tile get ids 8 apart inside a group of 64 (id = 64 q + 8 m + r): workgroups are dealt to
// the 8 XCDs round-robin by id, so the eight mirrors of a tile sit on ONE XCD next to each other in its queue and share the tile's
// weight rows through that XCD's L2 (measured: 84 % L2 hits on the weight stream, 87.5 % = 7 of 8 is the ideal).
//   FWD: src = cubemap level [6,R,R,3], scale = pixel_area / 4 per texel:
//            dst[o] = sum_i g(o,i) (src[i] scale[i]) / sum_i g(o,i) scale[i]
//   BWD: src = d loss / d out [6,R,R,3], scale = 1 / wsum per texel, out_scale = pixel_area / 4:
//            dst[i] = out_scale[i] sum_o g(o,i) (src[o] scale[o])
// (the reference multiplies g by area / 4 per pair, cubemap.cu:286-289; here the factor is applied once per staged source / once
// per result: a different rounding order of the same products, 1e-7 relative.)
struct MirrorMap { int face; int fx, fy; };
__device__ __forceinline__ MirrorMap mirror_map(int s, int m)          // the reflection m as seen from face s (see mirror_texel)
{
    const int pair = s >> 1;
    const int ax = pair == 0 ? 2 : 0;
    const int ay = pair == 1 ? 2 : 1;
    MirrorMap r;
    r.fx = (m >> ax) & 1; r.fy = (m >> ay) & 1; r.face = s;
    if ((m >> pair) & 1) {
        r.face = s ^ 1;
        if (pair == 1) r.fy ^= 1; else r.fx ^= 1;
    }
    return r;
}

template <bool BWD, bool SHARED_ROWS>
__device__ __forceinline__ void
tile_apply_body(int wg /*workgroup of this level's launch*/, int R, int E, int bw, int nb, int K, const float* __restrict__ src,
                const float* __restrict__ scale,
                  const float* __restrict__ out_scale, const float* __restrict__ bounds, const int4* __restrict__ tiles,
                  const int4* __restrict__ seg /*[nt][6]: x0, y0, rows, pitch*/, const int64_t* __restrict__ row_begin,
                  const int32_t* __restrict__ row_cnt, const int32_t* __restrict__ desc, const float* __restrict__ weights,
                  float* __restrict__ dst, int tile_begin, int tile_end)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_src[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = wave / K, part = wave - b * K;                     // block of the tile, share of its chunks
    int tile, m;
    if (SHARED_ROWS) {
        const int q = wg >> 6, r = wg & 63;
        m = r >> 3;
        tile = tile_begin + q * 8 + (r & 7);
    } else {
        m = 0;
        tile = tile_begin + wg;
    }
    if (tile >= tile_end) return;                                    // (the whole workgroup: uniform)
    const int4 tl = tiles[tile];
    const int own_s = tl.x, own_x = tl.y + 8 * (b % bw) + (lane & 7), own_y = tl.z + 8 * (b / bw) + (lane >> 3);
    const int t = (own_s * R + own_y) * R + own_x;                   // table texel (the octant's, when rows are shared)
    gs_f2 acc01 = { 0.0f, 0.0f }, acc23 = { 0.0f, 0.0f };
    for (int face = 0; face < 6; ++face) {
        const int4 sg = seg[tile * 6 + face];                        // x0, y0, rows of the region, pitch (0 = nothing on this face)
        const int pitch = sg.w;
        if (pitch == 0) continue;                                    // uniform over the workgroup
        __syncthreads();                                             // the previous face's readers are done
        {
            // stage the source rectangle: element e -> (yy, xx) = divmod(e, pitch); exact through the float reciprocal because
            // (e + 0.5) / pitch stays 0.5 / 128 away from every integer while the product's rounding error is < 2e-4
            const MirrorMap mm = mirror_map(face, m);
            const int total = sg.z * pitch;
            const float inv_pitch = 1.0f / (float)pitch;
            const int face_base = mm.face * R;
            for (int e0 = threadIdx.x; e0 < total; e0 += 4 * 1024) {
                // four elements per thread and round, their loads issued together from clamped indices (no branch around a load:
                // the round costs ONE memory latency instead of four)
                struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
                F3 p[4]; float a[4]; bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = min(e0 + u * 1024, total - 1);
                    const int yy = (int)(((float)e + 0.5f) * inv_pitch);
                    const int xx = e - yy * pitch;
                    const int x = sg.x + xx, y = sg.y + yy;
                    ok[u] = x >= 0 && x < R && y >= 0 && y < R;
                    const int xc = min(max(x, 0), R - 1), yc = min(max(y, 0), R - 1);
                    const int xm = mm.fx ? R - 1 - xc : xc, ym = mm.fy ? R - 1 - yc : yc;
                    const int idx = (face_base + ym) * R + xm;
#ifdef GS_TILE_EXP_NOSTAGE
                    p[u] = F3{ (float)idx, 1.0f, 2.0f }; a[u] = 1.0f;
#else
                    p[u] = *reinterpret_cast<const F3*>(src + (size_t)idx * 3);
                    a[u] = scale[idx];
#endif
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + u * 1024;
                    if (e < total) {
                        const float s = ok[u] ? a[u] : 0.0f;         // outside the face: finite zeros behind zero weights
                        s_src[e] = make_float4(p[u].x * s, p[u].y * s, p[u].z * s, BWD ? 0.0f : s);
                    }
                }
            }
        }
        __syncthreads();
        const int slot = (tile * 6 + face) * nb + b;
        const int n = row_cnt[slot];                                 // multiple of GS_TILE_ROW_PAD
        const int nc = n >> 3;                                       // chunks of the block
        const int nj = nc > part ? (nc - part + K - 1) / K : 0;      // ... of which this wave takes part, part + K, ...
        if (nj > 0) {
            const LaneBox bx = lane_box(bounds, t, face, E, R);
            const int a_l = bx.any ? ((bx.ymin - sg.y) * pitch + (bx.xmin - sg.x)) : 0;
            const int64_t r0 = row_begin[slot];
            const int4* dp = reinterpret_cast<const int4*>(desc + r0);
            // EIGHT register sets of 8 rows each, used round-robin: the loads of chunk j + 7 are issued before chunk j is consumed and
            // nothing is copied between the sets, so the wait in front of a chunk covers ITS loads only; seven younger chunks (56 rows,
            // 14 KB per wave) stay in flight.  That depth is what the kernel lives on: every weight line is fetched from HBM by
            // whichever of the eight mirror workgroups asks first and the other seven wait for the same fill, loads return in order,
            // so EVERY chunk sees the full HBM latency (~2 us under load) -- with three chunks in flight a wave advanced one chunk
            // per ~1500 cycles.  Weights are stored in row PAIRS ([row / 2][lane][2]): one 8-byte load per lane lands in an aligned
            // register pair whose halves feed the packed FMAs directly.  A chunk index past the end is clamped to the last chunk
            // (loaded again, not consumed): no branch around a load.  The row descriptors (scalar loads) run two chunks ahead.
            const char* lds_b = reinterpret_cast<const char*>(s_src) + a_l * 16;
            // weight rows: UNIFORM base pointer (an SGPR pair, advanced with scalar adds) + the lane's 32-bit byte offset + an immediate
            // per row pair -- `global_load_dwordx2 v, v_lane8, s[base] offset:512 u` -- instead of four 64-bit vector additions per chunk
            const gs_f2* wbase = reinterpret_cast<const gs_f2*>(weights) + (r0 >> 1) * 64;
            const unsigned lane_u = (unsigned)lane;
            // (round 5, measured and removed: the row descriptors through the VECTOR memory path -- a scalar load shares lgkmcnt with
            //  the LDS reads, so every chunk starts behind `s_waitcnt lgkmcnt(0)` -- 0.93 + 0.95 ms against 0.78 + 0.80: two more
            //  vector loads per chunk in a loop whose bound is the return order of its vector loads.  The scalar-base form of the
            //  weight loads above removes 25 of 35 64-bit vector additions and 60 scalar instructions: no change either.)
#ifdef GS_TILE_EXP_NOW
#define GS_TILE_WLOAD(P) (gs_f2{ 1.0f, (float)kk })
#elif defined(GS_TILE_WLOAD_SC1)
#define GS_TILE_WLOAD(P) (__builtin_bit_cast(gs_f2, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(P), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
#else
#define GS_TILE_WLOAD(P) (SHARED_ROWS ? *(P) : __builtin_nontemporal_load(P))
#endif
#ifdef GS_TILE_EXP_NODESC
#define GS_TILE_DLOAD(D0, D1, KK) do { D0 = make_int4((KK) * 16, (KK) * 16 + 16, (KK) * 16 + 32, (KK) * 16 + 48); D1 = make_int4((KK) * 16 + 64, (KK) * 16 + 80, (KK) * 16 + 96, (KK) * 16 + 112); } while (0)
#else
#define GS_TILE_DLOAD(D0, D1, KK) do { D0 = dp[(KK) >> 2]; D1 = dp[((KK) >> 2) + 1]; } while (0)
#endif
#define GS_CHUNK_ROW(J) ((min((J), nj - 1) * K + part) * 8)
#define GS_ROWS_LOAD(W, J)                                                                                          \
            do {                                                                                                    \
                const int kk = GS_CHUNK_ROW(J);                                                                     \
                const gs_f2* wrow = wbase + (size_t)(kk >> 1) * 64;                                                 \
                _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                       \
                    W[u] = GS_TILE_WLOAD(wrow + (lane_u + (unsigned)u * 64u));                                      \
            } while (0)
#define GS_DESC_LOAD(D0, D1, J)                                                                                     \
            do { const int kk = GS_CHUNK_ROW(J); GS_TILE_DLOAD(D0, D1, kk); } while (0)
            // acc += w * v for one row as TWO packed FMAs.  The weight of an even row is the low half of its register pair, that of
            // an odd row the high half; `op_sel` broadcasts either half to both result lanes (the compiler only knows the
            // low-half form and copies every odd-row weight into a fresh even register first: 8 v_mov per chunk and a full
            // wait in front of them).
#define GS_ROW_FMA(WPAIR, V, SEL)                                                                                   \
            do {                                                                                                    \
                const gs_f2 vxy = __builtin_shufflevector(V, V, 0, 1), vzw = __builtin_shufflevector(V, V, 2, 3);   \
                asm("v_pk_fma_f32 %0, %1, %2, %0 " SEL : "+v"(acc01) : "v"(WPAIR), "v"(vxy));                       \
                asm("v_pk_fma_f32 %0, %1, %2, %0 " SEL : "+v"(acc23) : "v"(WPAIR), "v"(vzw));                       \
            } while (0)
#define GS_SEL_LO "op_sel_hi:[0,1,1]"
#define GS_SEL_HI "op_sel:[1,0,0] op_sel_hi:[1,1,1]"
#ifdef GS_TILE_EXP_NOLDS
#define GS_LDS4(OFF) (gs_f4{ (float)(OFF), 1.0f, 2.0f, 3.0f })
#else
#define GS_LDS4(OFF) (*reinterpret_cast<const gs_f4*>(lds_b + (OFF)))
#endif
#define GS_ROWS_USE(W, D0, D1)                                                                                      \
            do {                                                                                                    \
                const gs_f4 t0 = GS_LDS4(D0.x), t1 = GS_LDS4(D0.y), t2 = GS_LDS4(D0.z), t3 = GS_LDS4(D0.w),         \
                            t4 = GS_LDS4(D1.x), t5 = GS_LDS4(D1.y), t6 = GS_LDS4(D1.z), t7 = GS_LDS4(D1.w);         \
                GS_ROW_FMA(W[0], t0, GS_SEL_LO); GS_ROW_FMA(W[0], t1, GS_SEL_HI);                                   \
                GS_ROW_FMA(W[1], t2, GS_SEL_LO); GS_ROW_FMA(W[1], t3, GS_SEL_HI);                                   \
                GS_ROW_FMA(W[2], t4, GS_SEL_LO); GS_ROW_FMA(W[2], t5, GS_SEL_HI);                                   \
                GS_ROW_FMA(W[3], t6, GS_SEL_LO); GS_ROW_FMA(W[3], t7, GS_SEL_HI);                                   \
            } while (0)
            // one step of the rotation: prefetch into the set that was consumed last, consume set WC with descriptor set (DA, DB),
            // then refill that descriptor set for two chunks later
#ifndef GS_DESC_AHEAD
#define GS_DESC_AHEAD 4                      // descriptor sets in flight (scalar loads): 2 until round 6
#endif
#define GS_STEP(WP, JP, WC, JC, DA, DB)                                                                             \
            do {                                                                                                    \
                GS_ROWS_LOAD(WP, JP);                                                                               \
                if ((JC) < nj) GS_ROWS_USE(WC, DA, DB);                                                             \
                GS_DESC_LOAD(DA, DB, (JC) + GS_DESC_AHEAD);                                                         \
            } while (0)
#if GS_TILE_DEPTH == 8
            gs_f2 w0[4], w1[4], w2[4], w3[4], w4[4], w5[4], w6[4], w7[4];
            GS_ROWS_LOAD(w0, 0); GS_ROWS_LOAD(w1, 1); GS_ROWS_LOAD(w2, 2); GS_ROWS_LOAD(w3, 3);
            GS_ROWS_LOAD(w4, 4); GS_ROWS_LOAD(w5, 5); GS_ROWS_LOAD(w6, 6);
#if GS_DESC_AHEAD == 2
            int4 a0, a1, b0, b1;
            GS_DESC_LOAD(a0, a1, 0); GS_DESC_LOAD(b0, b1, 1);
            for (int j = 0; j < nj; j += 8) {
                GS_STEP(w7, j + 7, w0, j + 0, a0, a1);
                GS_STEP(w0, j + 8, w1, j + 1, b0, b1);
                GS_STEP(w1, j + 9, w2, j + 2, a0, a1);
                GS_STEP(w2, j + 10, w3, j + 3, b0, b1);
                GS_STEP(w3, j + 11, w4, j + 4, a0, a1);
                GS_STEP(w4, j + 12, w5, j + 5, b0, b1);
                GS_STEP(w5, j + 13, w6, j + 6, a0, a1);
                GS_STEP(w6, j + 14, w7, j + 7, b0, b1);
            }
#elif GS_DESC_AHEAD == 4
            // The row descriptors are SCALAR loads of 32 bytes per chunk from a list that every wave walks once: every second chunk
            // misses the scalar cache, and with two sets in flight the miss (an L2 round trip) was exposed in front of most chunks --
            // with the weight loads compiled out the kernel still took 70 % of its time (profiles/r06_prefilter_removal.txt).
            int4 a0, a1, b0, b1, c0, c1, d0, d1;
            GS_DESC_LOAD(a0, a1, 0); GS_DESC_LOAD(b0, b1, 1); GS_DESC_LOAD(c0, c1, 2); GS_DESC_LOAD(d0, d1, 3);
            for (int j = 0; j < nj; j += 8) {
                GS_STEP(w7, j + 7, w0, j + 0, a0, a1);
                GS_STEP(w0, j + 8, w1, j + 1, b0, b1);
                GS_STEP(w1, j + 9, w2, j + 2, c0, c1);
                GS_STEP(w2, j + 10, w3, j + 3, d0, d1);
                GS_STEP(w3, j + 11, w4, j + 4, a0, a1);
                GS_STEP(w4, j + 12, w5, j + 5, b0, b1);
                GS_STEP(w5, j + 13, w6, j + 6, c0, c1);
                GS_STEP(w6, j + 14, w7, j + 7, d0, d1);
            }
#else
            int4 a0, a1, b0, b1, c0, c1, d0, d1, e0, e1, f0, f1, g0, g1, h0, h1;
            GS_DESC_LOAD(a0, a1, 0); GS_DESC_LOAD(b0, b1, 1); GS_DESC_LOAD(c0, c1, 2); GS_DESC_LOAD(d0, d1, 3);
            GS_DESC_LOAD(e0, e1, 4); GS_DESC_LOAD(f0, f1, 5); GS_DESC_LOAD(g0, g1, 6); GS_DESC_LOAD(h0, h1, 7);
            for (int j = 0; j < nj; j += 8) {
                GS_STEP(w7, j + 7, w0, j + 0, a0, a1);
                GS_STEP(w0, j + 8, w1, j + 1, b0, b1);
                GS_STEP(w1, j + 9, w2, j + 2, c0, c1);
                GS_STEP(w2, j + 10, w3, j + 3, d0, d1);
                GS_STEP(w3, j + 11, w4, j + 4, e0, e1);
                GS_STEP(w4, j + 12, w5, j + 5, f0, f1);
                GS_STEP(w5, j + 13, w6, j + 6, g0, g1);
                GS_STEP(w6, j + 14, w7, j + 7, h0, h1);
            }
#endif
#else
            gs_f2 w0[4], w1[4], w2[4], w3[4];
            GS_ROWS_LOAD(w0, 0); GS_ROWS_LOAD(w1, 1); GS_ROWS_LOAD(w2, 2);
#if GS_DESC_AHEAD == 2
            int4 a0, a1, b0, b1;
            GS_DESC_LOAD(a0, a1, 0); GS_DESC_LOAD(b0, b1, 1);
            for (int j = 0; j < nj; j += 4) {
                GS_STEP(w3, j + 3, w0, j + 0, a0, a1);
                GS_STEP(w0, j + 4, w1, j + 1, b0, b1);
                GS_STEP(w1, j + 5, w2, j + 2, a0, a1);
                GS_STEP(w2, j + 6, w3, j + 3, b0, b1);
            }
#elif GS_DESC_AHEAD == 4
            int4 a0, a1, b0, b1, c0, c1, d0, d1;
            GS_DESC_LOAD(a0, a1, 0); GS_DESC_LOAD(b0, b1, 1); GS_DESC_LOAD(c0, c1, 2); GS_DESC_LOAD(d0, d1, 3);
            for (int j = 0; j < nj; j += 4) {
                GS_STEP(w3, j + 3, w0, j + 0, a0, a1);
                GS_STEP(w0, j + 4, w1, j + 1, b0, b1);
                GS_STEP(w1, j + 5, w2, j + 2, c0, c1);
                GS_STEP(w2, j + 6, w3, j + 3, d0, d1);
            }
#else
            int4 a0, a1, b0, b1, c0, c1, d0, d1, e0, e1, f0, f1, g0, g1, h0, h1;
            GS_DESC_LOAD(a0, a1, 0); GS_DESC_LOAD(b0, b1, 1); GS_DESC_LOAD(c0, c1, 2); GS_DESC_LOAD(d0, d1, 3);
            GS_DESC_LOAD(e0, e1, 4); GS_DESC_LOAD(f0, f1, 5); GS_DESC_LOAD(g0, g1, 6); GS_DESC_LOAD(h0, h1, 7);
            for (int j = 0; j < nj; j += 8) {
                GS_STEP(w3, j + 3, w0, j + 0, a0, a1);
                GS_STEP(w0, j + 4, w1, j + 1, b0, b1);
                GS_STEP(w1, j + 5, w2, j + 2, c0, c1);
                GS_STEP(w2, j + 6, w3, j + 3, d0, d1);
                GS_STEP(w3, j + 7, w0, j + 4, e0, e1);
                GS_STEP(w0, j + 8, w1, j + 5, f0, f1);
                GS_STEP(w1, j + 9, w2, j + 6, g0, g1);
                GS_STEP(w2, j + 10, w3, j + 7, h0, h1);
            }
#endif
#endif
#undef GS_STEP
#undef GS_DESC_LOAD
#undef GS_CHUNK_ROW
#undef GS_TILE_DLOAD
#undef GS_TILE_WLOAD
#undef GS_ROWS_LOAD
#undef GS_ROWS_USE
#undef GS_ROW_FMA
#undef GS_SEL_LO
#undef GS_SEL_HI
#undef GS_LDS4
        }
    }
    // The staged texels are float4 so that a row costs ONE ds_read_b128 (4 LDS cycles; the 12-byte ds_read_b96 takes 8).  The
    // fourth component carries pixel_area / 4 in the forward -- acc23.y is then the sum of this texel's pair weights, in the order
    // of the colour sums it normalises -- and 0 in the backward.
    if (K > 1) {                                                     // the K shares of a block meet in LDS, summed in a fixed order
        __syncthreads();
        s_src[wave * 64 + lane] = make_float4(acc01.x, acc01.y, acc23.x, acc23.y);
        __syncthreads();
        if (part != 0) return;
        float4 sum = s_src[wave * 64 + lane];
        for (int k = 1; k < K; ++k) {
            const float4 o = s_src[(wave + k) * 64 + lane];
            sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
        }
        acc01.x = sum.x; acc01.y = sum.y; acc23.x = sum.z; acc23.y = sum.w;
    }
    const int to = mirror_texel(own_s, own_x, own_y, m, R);
    float* q = dst + (size_t)to * 3;
    if (BWD) {
        const float a = out_scale[to];
        q[0] = acc01.x * a; q[1] = acc01.y * a; q[2] = acc23.x * a;
    } else {
        q[0] = acc01.x / acc23.y; q[1] = acc01.y / acc23.y; q[2] = acc23.x / acc23.y;
    }
}

template <bool BWD, bool SHARED_ROWS>
__global__ void __launch_bounds__(1024)
tile_apply_kernel(int R, int E, int bw, int nb, int K, const float* __restrict__ src, const float* __restrict__ scale,
                  const float* __restrict__ out_scale, const float* __restrict__ bounds, const int4* __restrict__ tiles,
                  const int4* __restrict__ seg, const int64_t* __restrict__ row_begin, const int32_t* __restrict__ row_cnt,
                  const int32_t* __restrict__ desc, const float* __restrict__ weights, float* __restrict__ dst, int tile_begin, int tile_end)
{
    tile_apply_body<BWD, SHARED_ROWS>((int)blockIdx.x, R, E, bw, nb, K, src, scale, out_scale, bounds, tiles, seg, row_begin, row_cnt, desc,
                                      weights, dst, tile_begin, tile_end);
}

// ALL levels of the pyramid in ONE launch (round 6).  As six launches the three levels below 128^2 -- 24 + 96 + 192 workgroups that
// each walk a long row list -- held the GPU for 0.13 ms per direction with a tenth of its CUs busy, and every launch paid its own
// fill and drain (profiles/r05_step_boundary.txt: 0.207 + 0.300 + 0.150 + 0.060 + 0.037 + 0.032 ms).  Here a workgroup finds its
// level from a prefix table in the kernel arguments (<= GS_TILE_MAX_MULTI entries, uniform scalar compares) and runs the body above
// with that level's geometry; the levels do not depend on each other (each reads its own mip / cotangent), the dynamic LDS is the
// largest level's.  Same arithmetic per workgroup: bit-identical to the per-level launches.
#define GS_TILE_MAX_MULTI 8
struct TileLevelDev {
    int R, E, bw, nb, K, tile_begin, tile_end, wg_begin;
    const float* src; const float* scale; const float* out_scale; const float* bounds;
    const int4* tiles; const int4* seg; const int64_t* row_begin; const int32_t* row_cnt; const int32_t* desc;
    const float* weights; float* dst;
};
struct TileMultiArgs { int n; TileLevelDev lv[GS_TILE_MAX_MULTI]; };

template <bool BWD, bool SHARED_ROWS>
__global__ void __launch_bounds__(1024)
tile_apply_multi_kernel(const TileMultiArgs a)
{
    int l = 0;
    for (int k = 1; k < GS_TILE_MAX_MULTI; ++k) if (k < a.n && (int)blockIdx.x >= a.lv[k].wg_begin) l = k;
    const TileLevelDev& v = a.lv[l];
    tile_apply_body<BWD, SHARED_ROWS>((int)blockIdx.x - v.wg_begin, v.R, v.E, v.bw, v.nb, v.K, v.src, v.scale, v.out_scale, v.bounds, v.tiles,
                                      v.seg, v.row_begin, v.row_cnt, v.desc, v.weights, v.dst, v.tile_begin, v.tile_end);
}

extern "C" int gs_specular_tiles_apply_multi(int n_levels, const GsTileLevel* levels, int backward, void* stream)
{
    GS_CHECK_ARG(n_levels >= 1 && n_levels <= GS_TILE_MAX_MULTI && levels != nullptr, "1..8 levels");
    TileMultiArgs a;
    a.n = 0;
    size_t lds_bytes = 16 * 64 * 16;                                 // the partial sums of the 16 waves
    int grid = 0;
    const int n_mirrors = levels[0].n_mirrors;
    for (int i = 0; i < n_levels; ++i) {
        const GsTileLevel& g = levels[i];
        GS_CHECK_ARG(tile_geometry_ok(g.R, g.bw, g.nb) && (16 % g.nb) == 0 && (g.n_mirrors == 1 || g.n_mirrors == 8),
                     "R must be a multiple of 16, nb in {1, 2, 4, 8, 16}, n_mirrors 1 or 8");
        GS_CHECK_ARG(g.n_mirrors == n_mirrors, "every level of a merged launch must use the same n_mirrors");
        GS_CHECK_ARG(g.src && g.scale && (g.out_scale || !backward) && g.bounds && g.tiles && g.segments && g.row_begin && g.row_counts && g.desc
                     && g.weights && g.dst, "null argument");
        GS_CHECK_ARG(g.tile_begin >= 0 && g.tile_begin <= g.tile_end && g.lds_bytes <= 160 * 1024, "bad tile range / LDS size");
        if (g.tile_begin == g.tile_end) continue;
        const int nt = g.tile_end - g.tile_begin;
        TileLevelDev& v = a.lv[a.n++];
        v.R = g.R; v.E = backward ? g.margin : 0; v.bw = g.bw; v.nb = g.nb; v.K = 16 / g.nb; v.tile_begin = g.tile_begin; v.tile_end = g.tile_end;
        v.wg_begin = grid;
        v.src = g.src; v.scale = g.scale; v.out_scale = g.out_scale; v.bounds = g.bounds; v.tiles = (const int4*)g.tiles;
        v.seg = (const int4*)g.segments; v.row_begin = g.row_begin; v.row_cnt = g.row_counts; v.desc = g.desc; v.weights = g.weights; v.dst = g.dst;
        grid += n_mirrors == 8 ? gs_cdiv(nt, 8) * 64 : nt;
        if (g.lds_bytes > lds_bytes) lds_bytes = g.lds_bytes;
    }
    if (a.n == 0) return GS_OK;
    const hipStream_t s = (hipStream_t)stream;
#define GS_TILE_MULTI_LAUNCH(BW, SH)                                                                                                   \
    do {                                                                                                                               \
        GS_CHECK_HIP(hipFuncSetAttribute((const void*)tile_apply_multi_kernel<BW, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        hipLaunchKernelGGL((tile_apply_multi_kernel<BW, SH>), dim3(grid), dim3(1024), lds_bytes, s, a);                                \
    } while (0)
    if (n_mirrors == 8) { if (backward) GS_TILE_MULTI_LAUNCH(true, true); else GS_TILE_MULTI_LAUNCH(false, true); }
    else { if (backward) GS_TILE_MULTI_LAUNCH(true, false); else GS_TILE_MULTI_LAUNCH(false, false); }
#undef GS_TILE_MULTI_LAUNCH
    GS_CHECK_LAUNCH();
    return GS_OK;
}

extern "C" int gs_specular_tiles_apply(int R, int backward, int n_mirrors, int margin, int bw, int nb, const float* src,
                                       const float* scale, const float* out_scale, const float* bounds, const int32_t* tiles,
                                       const int32_t* segments, const int64_t* row_begin, const int32_t* row_counts, const int32_t* desc,
                                       const float* weights, float* dst, int tile_begin, int tile_end, size_t lds_bytes, void* stream)
{
    GS_CHECK_ARG(tile_geometry_ok(R, bw, nb) && (16 % nb) == 0 && (n_mirrors == 1 || n_mirrors == 8),
                 "R must be a multiple of 16, nb in {1, 2, 4, 8, 16}, n_mirrors 1 or 8");
    GS_CHECK_ARG(src && scale && (out_scale || !backward) && bounds && tiles && segments && row_begin && row_counts && desc && weights && dst,
                 "null argument");
    GS_CHECK_ARG(tile_begin >= 0 && tile_begin <= tile_end && lds_bytes <= 160 * 1024, "bad tile range / LDS size");
    if (tile_begin == tile_end) return GS_OK;
    const int nt = tile_end - tile_begin;
    const int K = 16 / nb;
    if (lds_bytes < 16 * 64 * 16) lds_bytes = 16 * 64 * 16;          // the partial sums of the 16 waves
    const hipStream_t s = (hipStream_t)stream;
#define GS_TILE_LAUNCH(BW, SH, GRID)                                                                                                   \
    do {                                                                                                                               \
        /* set on EVERY launch (as gs_shade.hip does): the attribute is per device, a process may drive several GPUs / threads */     \
        GS_CHECK_HIP(hipFuncSetAttribute((const void*)tile_apply_kernel<BW, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        hipLaunchKernelGGL((tile_apply_kernel<BW, SH>), dim3(GRID), dim3(1024), lds_bytes, s, R, (BW) ? margin : 0, bw, nb, K, src, scale,  \
                           out_scale, bounds, (const int4*)tiles, (const int4*)segments, row_begin, row_counts, desc, weights, dst,   \
                           tile_begin, tile_end);                                                                                      \
    } while (0)
    if (n_mirrors == 8) {
        const int grid = gs_cdiv(nt, 8) * 64;
        if (backward) GS_TILE_LAUNCH(true, true, grid); else GS_TILE_LAUNCH(false, true, grid);
    } else {
        if (backward) GS_TILE_LAUNCH(true, false, nt); else GS_TILE_LAUNCH(false, false, nt);
    }
#undef GS_TILE_LAUNCH
    GS_CHECK_LAUNCH();
    return GS_OK;
}
