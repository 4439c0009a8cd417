// gs_raster.hip -- A5 / A6: front-to-back alpha compositing of per-tile depth-sorted Gaussian lists and
// its stored-state backward (gsplat 1.4 `rasterize_to_pixels` fwd/bwd as reached from
// rfstudio/model/gsplat.py:334-355; constants and loop semantics in SURVEY.md section 8a rows A5/A6).
//
// CDNA4 mapping (not the upstream 256-thread-block / shared-memory-batch design):
//   * STREAM: after the sort, one pass gathers every intersection's geometry (+ colour for D<=3) into a
//     record stream laid out in SORTED order (3 x float4 per intersection, SoA).  The compositor then never
//     chases an index: a wave reads 64 consecutive records with three 1-KiB coalesced loads, and the next
//     batch is prefetched into registers while the current one is composited.
//   * a 16x16 tile is split into four 8x8 QUADRANTS, one wave64 each; a wave never synchronises with the
//     other three -- no LDS staging, no __syncthreads, each wave stops as soon as ITS 64 pixels are opaque;
//   * lane l of a batch tests record l's {alpha >= 1/255} extent (precomputed in the stream) against the
//     quadrant rectangle; a 64-bit ballot compacts the survivors, which are broadcast lane->SGPR with
//     v_readlane and evaluated by all 64 pixel lanes.  Exact: a culled Gaussian has alpha < 1/255 at every
//     pixel centre of the quadrant and the reference semantics would skip it anyway.
//   * tiles are launched longest-list-first (LPT order from a single-block bucketing kernel) so that the
//     silhouette tiles, whose lists are 4x the mean, do not form the tail of the launch;
//   * backward: same walk back-to-front from the wave's max(last_ids); per-Gaussian partials are summed
//     across the wave with DPP row operations and ONE lane issues the fp32 atomics.
// Both kernels are FP32-VALU / transcendental bound (one v_exp_f32 per evaluated pair), not HBM bound.
#include "gs_common.h"
#include "gs_tone.h"
#include <stdlib.h>

#pragma clang fp contract(off)   // sigma / compositing are spelled with explicit fmaf (bit-exact vs oracle)

#define GS_TILE 16
#define GS_ALPHA_MIN (1.0f / 255.0f)
#ifndef GS_BWD_ROW_COMMIT
#define GS_BWD_ROW_COMMIT 0      // 1 = every DPP row commits its own partial sums (no cross-row exchange): measured 2.2x SLOWER --
                                 // the kernel sits at the memory-side atomic request rate (~4 requests per survivor already)
#endif

// Diagnostic build only (scripts/raster_stats.py compiles this file with -DGS_RASTER_STATS into a scratch .so):
// counts wave-batches, ballot survivors and evaluated/valid lane-pairs.  Never defined in the product library.
#ifdef GS_RASTER_STATS
__device__ unsigned long long g_raster_stats[8];
extern "C" int gs_raster_stats_read(unsigned long long* host8, int reset)
{
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_raster_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_raster_stats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
// second bank (round 4): what the UNUSED candidate slots of the backward walk are.  [0] candidates popped by the backward walk, [1] its
// dense batches, [2] popped but rejected because the pixel had terminated in front of the record (idx > last_ids), [3] popped but
// rejected by sigma < 0 / alpha < 1/255 (the slack of the ellipse masks), [4] forward: candidates popped, [5] forward: dense batches,
// [6] backward: sum over dense batches of the LONGEST per-pixel list (lock-step length), [7] backward: sum of the reduction's trips
__device__ unsigned long long g_raster_stats2[8];
extern "C" int gs_raster_stats2_read(unsigned long long* host8, int reset)
{
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_raster_stats2), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_raster_stats2), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
// third bank (round 4): walk trips of the cull-log backward by the number of lanes that still hold a candidate: [k] = trips with
// 8 k < lanes <= 8 (k + 1)
__device__ unsigned long long g_raster_stats3[8];
extern "C" int gs_raster_stats3_read(unsigned long long* host8, int reset)
{
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_raster_stats3), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_raster_stats3), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
// per-block timeline (diagnostic, -DGS_RASTER_PHASES): [3 * b] = first wave start, [3 * b + 1] = last wave end (100 MHz wall clock),
// [3 * b + 2] = list length of the tile
#define GS_TL_MAX 16384
__device__ unsigned long long g_raster_tl[3 * GS_TL_MAX];
extern "C" int gs_raster_timeline_read(unsigned long long* host, int n_blocks, int reset)
{
    if (n_blocks > GS_TL_MAX) return -1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_raster_tl), sizeof(unsigned long long) * 3 * n_blocks) != hipSuccess) return -1;
    if (reset) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_raster_tl)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * 3 * GS_TL_MAX) != hipSuccess) return -1; }
    return 0;
}
#ifdef GS_RASTER_PHASES
#define GS_TL_BEGIN(len) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < GS_TL_MAX) { const unsigned long long _t = wall_clock64(); \
    unsigned long long* _e = g_raster_tl + 3 * blockIdx.x; if (threadIdx.x == 0) { _e[2] = (unsigned long long)(len); } \
    atomicMax(_e, ~_t); } } while (0)
#define GS_TL_END() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < GS_TL_MAX) atomicMax(g_raster_tl + 3 * blockIdx.x + 1, wall_clock64()); } while (0)
#endif
#define GS_STAT(i, v) do { const unsigned long long _sv = (unsigned long long)(v); if ((threadIdx.x & 63) == 0) atomicAdd(&g_raster_stats[i], _sv); } while (0)
#define GS_STAT_ALL(i, v) atomicAdd(&g_raster_stats[i], (unsigned long long)(v))      /* every active lane adds */
#define GS_STAT2(i, v) do { const unsigned long long _sv = (unsigned long long)(v); if ((threadIdx.x & 63) == 0) atomicAdd(&g_raster_stats2[i], _sv); } while (0)
#define GS_STAT2_ALL(i, v) atomicAdd(&g_raster_stats2[i], (unsigned long long)(v))
#ifdef GS_RASTER_PHASES            /* cycles of wave 0 of the LONGEST tile (block 0 in LPT order) per phase: overrides the counters above */
#undef GS_STAT
#undef GS_STAT_ALL
#undef GS_STAT2
#undef GS_STAT2_ALL
#define GS_STAT(i, v) do { } while (0)
#define GS_STAT_ALL(i, v) do { } while (0)
#define GS_STAT2(i, v) do { } while (0)
#define GS_STAT2_ALL(i, v) do { } while (0)
#define GS_PHASE_BEGIN() const long long _ph0 = (blockIdx.x == 0 && threadIdx.x == 0) ? (long long)__builtin_readcyclecounter() : 0
#define GS_PHASE_END(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_raster_stats[i], (unsigned long long)((long long)__builtin_readcyclecounter() - _ph0)); } while (0)
#endif
#else
#define GS_STAT(i, v) do { } while (0)
#define GS_STAT_ALL(i, v) do { } while (0)
#define GS_STAT2(i, v) do { } while (0)
#define GS_STAT2_ALL(i, v) do { } while (0)
#endif
#ifndef GS_TL_BEGIN
#define GS_TL_BEGIN(len) do { } while (0)
#define GS_TL_END() do { } while (0)
#endif
#ifndef GS_PHASE_BEGIN
#define GS_PHASE_BEGIN() do { } while (0)
#define GS_PHASE_END(i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------
// stream build: rec0 = {mx, my, 0.5a, b}, rec1 = {0.5c, opacity, hx, hy}, rec2 = {c0, c1, c2, bits(g)}
// (hx < 0 marks "can never reach alpha_min").  One thread per sorted intersection.
__global__ void __launch_bounds__(256)
build_stream_kernel(GsCount ic, int D, const int32_t* __restrict__ flatten_ids, const float* __restrict__ means2d,
                    const float* __restrict__ conics, const float* __restrict__ opacities,
                    const float* __restrict__ colors, float4* __restrict__ rec0, float4* __restrict__ rec1,
                    float4* __restrict__ rec2)
{
    const int n_isects = (int)gs_count(ic);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_isects) return;
    const int g = flatten_ids[i];
    const float2 m = *reinterpret_cast<const float2*>(means2d + 2 * (size_t)g);
    const float ca = conics[3 * (size_t)g], cb = conics[3 * (size_t)g + 1], cc = conics[3 * (size_t)g + 2];
    const float op = opacities[g];
    float hx = -1.0f, hy = -1.0f;
    if (!alpha_extent(ca, cb, cc, op, hx, hy)) { hx = -1.0f; hy = -1.0f; }
    rec0[i] = make_float4(m.x, m.y, 0.5f * ca, cb);
    rec1[i] = make_float4(0.5f * cc, op, hx, hy);
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (D <= 3) {
        c0 = colors[(size_t)g * D];
        if (D > 1) c1 = colors[(size_t)g * D + 1];
        if (D > 2) c2 = colors[(size_t)g * D + 2];
    }
    rec2[i] = make_float4(c0, c1, c2, __int_as_float(g));
}

// Same stream from the PACKED per-visible records written by gs_pack_visible (3 aligned 16-byte gathers per
// intersection instead of 8 scalar ones: the gather, not the 48-byte store, bounds this kernel).
__global__ void __launch_bounds__(256)
build_stream_packed_kernel(GsCount ic, const int32_t* __restrict__ flatten_ids, const float4* __restrict__ vis,
                           float4* __restrict__ rec0, float4* __restrict__ rec1, float4* __restrict__ rec2)
{
    const int n_isects = (int)gs_count(ic);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_isects) return;
    const int g = flatten_ids[i];
    const float4* v = vis + 4 * (size_t)g;          // one 64-byte record = one cache line per gathered Gaussian
    const float4 a = v[0], b = v[1];
    float4 c = v[2];
    c.w = __int_as_float(g);
    rec0[i] = a; rec1[i] = b; rec2[i] = c;
}

// per-visible packing: ONE 64-byte record per Gaussian {mx,my,0.5a,b | 0.5c,op,hx,hy | c0,c1,c2,- | pad} so that the
// stream build gathers one cache line per intersection (three separate arrays cost three sector fetches)
__global__ void __launch_bounds__(256)
pack_visible_kernel(int V, int D, const float* __restrict__ means2d, const float* __restrict__ conics,
                    const float* __restrict__ opacities, const float* __restrict__ colors,
                    float4* __restrict__ vis)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= V) return;
    const float2 m = *reinterpret_cast<const float2*>(means2d + 2 * (size_t)g);
    const float ca = conics[3 * (size_t)g], cb = conics[3 * (size_t)g + 1], cc = conics[3 * (size_t)g + 2];
    const float op = opacities[g];
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (D <= 3) {
        c0 = colors[(size_t)g * D];
        if (D > 1) c1 = colors[(size_t)g * D + 1];
        if (D > 2) c2 = colors[(size_t)g * D + 2];
    }
    gs_write_vis_record(vis + 4 * (size_t)g, m.x, m.y, ca, cb, cc, op, c0, c1, c2);
}

// ---------------------------------------------------------------------------------------------------
// LPT tile order: bucket tiles by floor(log2(count)) descending (single block).
// The backward's own tile order (longest quadrant log first: that wave bounds the block), built BY THE FORWARD without a launch of its
// own: at the end of a block its four waves leave their log lengths in LDS, and one thread files the tile in the list of its
// half-octave bucket (two atomics per tile).  The backward's block b finds its tile from the 64 bucket counts, longest bucket
// first.  bcount is zeroed by tile_order_kernel (gs_raster_prepare*), on the front stream.
struct BwdOrder { int32_t* bcount; int32_t* blist; };

__global__ void __launch_bounds__(1024)
tile_order_kernel(int n_tiles, GsCount ic, const int32_t* __restrict__ offsets, int32_t* __restrict__ order, BwdOrder bo)
{
    if (threadIdx.x < 64) bo.bcount[threadIdx.x] = 0;
    // longest list first, in buckets of HALF an octave (whole octaves until the end of round 4: the order inside a bucket is arbitrary)
    const int n_isects = (int)gs_count(ic);
    __shared__ int hist[64];
    __shared__ int base[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    auto bucket = [&](int t) {
        const int c = ((t == n_tiles - 1) ? n_isects : offsets[t + 1]) - offsets[t];
        if (c <= 0) return 63;                                       // empty: last
        const int l2 = 31 - __clz(c);
        const int half = (l2 > 0 && ((c >> (l2 - 1)) & 1)) ? 1 : 0;
        const int b = 2 * l2 + half;
        return 61 - (b > 61 ? 61 : b);
    };
    for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) atomicAdd(&hist[bucket(t)], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < 64; ++b) { base[b] = acc; acc += hist[b]; }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) order[atomicAdd(&base[bucket(t)], 1)] = t;
}

// ===================================================================================================
// PER-LANE LISTS over DENSE batches ("lanes" kernels).  The round-1 quadrant kernels (one wave per survivor; removed in round 5
// together with the single-batch forward -- `git show 7240552:geosplatting_amd/csrc/gs_raster.hip`) spent all 64 lanes on every
// survivor of the ballot cull although a surface splat at 2 M / 800^2 covers ~8 of the 64 pixels (7.8 valid lanes
// per survivor, 128 wave instructions each in the backward).  Here every PIXEL walks only its own candidates:
//   * CULL + COMPACT: a raw batch of 64 stream records is tested against the rectangle of the still-active pixels
//     (as before) and the ~20 survivors are appended, in stream order, to a wave-private LDS queue; a DENSE batch
//     of 64 surviving records is processed whenever the queue holds one (the per-batch work below is SIMD over the
//     records of a batch, so it is paid per 64 useful records instead of per 64 stream records);
//   * lane j of a dense batch turns record j's EXACT {alpha >= 1/255} ellipse (one interval per pixel row, with
//     slack) into a 64-bit pixel mask of the quadrant;
//   * a 64x64 bit-matrix transpose across the wave (6 exchange stages: ds_swizzle / ds_bpermute, no LDS storage)
//     hands lane p the 64-bit LIST of the records that can touch pixel p;
//   * lane p pops its list front to back (back to front in the backward), fetching "its" record from the LDS queue
//     -- one trip of the loop evaluates up to 64 DIFFERENT (pixel, Gaussian) pairs;
//   * backward: the 6+D per-Gaussian sums can no longer be reduced across the wave (every lane works on another
//     Gaussian): the walk stores two scalars per pair in an LDS pair buffer and the lanes then switch roles -- lane j
//     owns record j and sums its pairs (raster_bwd_lanes2_kernel below; until round 5 a variant with one ds_add_f64
//     accumulator row per record served D > 3) -- and commits once per (quadrant, Gaussian): 7 records x 9 values
//     per atomic instruction, the 9 lanes of a record falling into one 64-byte gradient record = one memory-side request.
// Per-pixel evaluation order and arithmetic are those of the quadrant kernels, so the forward is bit-identical.

template <int K>
__device__ __forceinline__ unsigned gs_lane_xor(unsigned v)
{
    if (K == 32) return (unsigned)__shfl_xor((int)v, 32, 64);
    return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (K << 10) | 0x1f);      // bit mode: lane ^ K inside 32 lanes
}

// out[lane p] bit j = in[lane j] bit p   (64 x 64 bits held as one 64-bit value per lane)
__device__ __forceinline__ unsigned long long gs_bit_transpose64(unsigned long long x, int lane)
{
    unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    {
        const bool up = (lane & 32) != 0;
        const unsigned recv = gs_lane_xor<32>(up ? lo : hi);
        if (up) lo = recv; else hi = recv;
    }
    // one exchange stage without divergence: the partner's word rotated right by K (upper lanes) or left by K (lower lanes;
    // = right by 32 - K) lines its wanted bits up with the positions this lane gives away; the bits that wrap around fall
    // outside the select mask.  2 VALU per word and stage (v_alignbit + v_bfi) -- the two-sided `up ? ... : ...` form made
    // the compiler emit both sides under exec masks, ~150 instructions per transposition.
#define GS_TR_STAGE(K, LOM)                                                                                        \
    {                                                                                                              \
        const bool up = (lane & K) != 0;                                                                           \
        const unsigned rot = up ? (unsigned)K : (unsigned)(32 - K), take = up ? LOM : ~LOM;                        \
        const unsigned rl = gs_lane_xor<K>(lo), rh = gs_lane_xor<K>(hi);                                           \
        lo = (__builtin_amdgcn_alignbit(rl, rl, rot) & take) | (lo & ~take);                                       \
        hi = (__builtin_amdgcn_alignbit(rh, rh, rot) & take) | (hi & ~take);                                       \
    }
    GS_TR_STAGE(16, 0x0000ffffu)
    GS_TR_STAGE(8, 0x00ff00ffu)
    GS_TR_STAGE(4, 0x0f0f0f0fu)
    GS_TR_STAGE(2, 0x33333333u)
    GS_TR_STAGE(1, 0x55555555u)
#undef GS_TR_STAGE
    return ((unsigned long long)hi << 32) | lo;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// active rectangle of a quadrant wave in QUADRANT pixel indices (0..7)
__device__ __forceinline__ void active_rect_i(unsigned long long act, int& xmin, int& xmax, int& ymin, int& ymax)
{
    unsigned cols = (unsigned)(act | (act >> 32));
    cols |= cols >> 16; cols |= cols >> 8; cols &= 0xffu;
    xmin = __builtin_ctz(cols); xmax = 31 - __builtin_clz(cols);
    ymin = __builtin_ctzll(act) >> 3; ymax = (63 - __builtin_clzll(act)) >> 3;
}

#ifndef GS_WALK_PAIRS
#define GS_WALK_PAIRS 1           // packed candidate PAIRS popped per lane and trip in the forward walk
#endif
#ifndef GS_LANES_EXACT_MASK
#define GS_LANES_EXACT_MASK 1
#endif

// 64-bit mask (bit y*8+x) of the quadrant pixels whose centre can lie inside the record's {alpha >= 1/255} set, clipped to
// the active rectangle.  A SUPERSET is all that is needed (every popped pair is tested exactly); the tighter it is, the
// shorter the per-pixel lists: the bounding rectangle of a thin diagonal ellipse holds twice the pixels of the ellipse.
//   sigma(dx, dy) = ha dx^2 + cb dx dy + hc dy^2 <= tau := log(255 op)     (ha = a/2, hc = c/2, dx = mx - px)
// per pixel row (fixed dy):  dx in [xc - w, xc + w],  xc = -cb dy / (2 ha),  w = sqrt(dy^2 (cb^2 - 4 ha hc) + 4 ha tau) / (2 ha).
__device__ __forceinline__ unsigned long long record_pixel_mask(bool ok, float mx, float my, float ha, float cb, float hc, float op,
                                                                int qx0, int qy0, int xmin, int xmax, int ymin, int ymax)
{
    // (the extents hx, hy of the stream record are not consulted: the per-row intervals bound the ellipse in both directions, and
    // the queue need not carry them -- 1 KB of LDS per wave)
    const float ox = (float)qx0 + 0.5f, oy = (float)qy0 + 0.5f;
    const float x0f = (float)xmin, x1f = (float)xmax;
    if (!ok) return 0ull;
    const int y0 = ymin, y1 = ymax;
    if (!GS_LANES_EXACT_MASK || !(ha > 1e-12f)) {                            // the active rectangle (degenerate conics)
        const unsigned colmask = ((2u << xmax) - 1u) & ~((1u << xmin) - 1u);
        const unsigned rep = colmask * 0x01010101u;
        const unsigned long long rows = (~0ull >> (8 * (7 - y1))) & (~0ull << (8 * y0));
        return ((((unsigned long long)rep) << 32) | rep) & rows;
    }
    // A superset with slack is all that is needed, so this runs on the raw hardware log2 / rcp / sqrt (1 ulp, no denormal or
    // range fix-ups: the precise sqrtf / division / logf expansions made this function 410 instructions per dense batch, a fifth
    // of the kernel), two pixel rows at a time in packed fp32.
    const float tau = __builtin_amdgcn_logf(255.0f * op) * 0.69314718f + 0.004f;     // slack >> rounding of sigma at a pixel
    const float inv2a = 0.5f * __builtin_amdgcn_rcpf(ha);
    const float k1 = cb * cb - 4.0f * ha * hc, k2 = 4.0f * ha * tau, k3 = cb * inv2a;
    const float mxo = mx - ox, dy0 = my - oy;
    unsigned lo = 0u, hi = 0u;
#pragma unroll
    for (int y = 0; y < 8; y += 2) {
        const v2f dy = (v2f)(dy0) - v2f{(float)y, (float)(y + 1)};
        const v2f disc = __builtin_elementwise_fma(dy * dy, (v2f)(k1), (v2f)(k2));
        const v2f root = v2f{__builtin_amdgcn_sqrtf(fmaxf(disc.x, 0.0f)), __builtin_amdgcn_sqrtf(fmaxf(disc.y, 0.0f))};
        const v2f w = __builtin_elementwise_fma(root, (v2f)(inv2a), (v2f)(0.01f));
        const v2f xc = __builtin_elementwise_fma(dy, (v2f)(k3), (v2f)(mxo));        // px - ox = mxo - dx,  dx centre = -k3 dy
        const v2f xlo = xc - w, xhi = xc + w;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int yy = y + u;
            const float xl = fmaxf(ceilf(u ? xlo.y : xlo.x), x0f), xh = fminf(floorf(u ? xhi.y : xhi.x), x1f);
            const bool on = (yy >= y0) && (yy <= y1) && ((u ? disc.y : disc.x) >= 0.0f) && (xl <= xh);
            const unsigned bits = on ? (((2u << (int)xh) - 1u) & (~0u << (int)xl)) : 0u;
            if (yy < 4) lo |= bits << (8 * yy); else hi |= bits << (8 * (yy - 4));
        }
    }
    return ((unsigned long long)hi << 32) | lo;
}

// index of the lowest set bit of a per-lane 64-bit list, which is cleared; an empty list yields 31 (a harmless in-range index)
// and stays empty.  v_ffbl_b32 returns -1 for zero, which C's ctz leaves undefined -- hence the two asm statements.
__device__ __forceinline__ int gs_pop_lowest(unsigned long long& list)
{
    unsigned lo = (unsigned)list, hi = (unsigned)(list >> 32), flo, fhi;
    asm("v_ffbl_b32 %0, %1" : "=v"(flo) : "v"(lo));
    asm("v_ffbl_b32 %0, %1" : "=v"(fhi) : "v"(hi));
    list &= list - 1ull;
    return (int)min(flo, fhi + 32u);
}

// sigma(dx, dy) = fma(ha dx, dx, fma(hc dy, dy, (cb dx) dy)) for ONE record from d = {dx, dy}, p = {ha dx, cb dx}.  The four scalar
// operations are inline assembly only so that the SLP vectoriser does not pair them with the other candidate's (it did, at the price
// of eight v_mov per trip to line the operands up); v_mul_f32 / v_fma_f32 are the instructions the compiler emits for the same C.
__device__ __forceinline__ float gs_sigma_xy(v2f d, v2f p, float hc)
{
    float t1, q, r, sg;
    asm("v_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(hc), "v"(d.y));
    asm("v_mul_f32 %0, %1, %2" : "=v"(q) : "v"(p.y), "v"(d.y));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(t1), "v"(d.y), "v"(q));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(sg) : "v"(p.x), "v"(d.x), "v"(r));
    return sg;
}

// gs_exp_neg for two candidates at once, for callers that discard results below 1/255: without the flush to zero below 2^-125
// and the upper clamp (sigma < 0 is rejected by the caller), which changes no result that survives the alpha >= 1/255 test --
// those have y >= -8.  The lower clamp stays: it turns an infinite or NaN y into -126 instead of a NaN that fminf would drop.
__device__ __forceinline__ v2f gs_exp_neg_live2(v2f sigma)
{
#pragma clang fp contract(off)
    v2f y = sigma * -1.44269504f;
    y.x = fmaxf(y.x, -126.0f); y.y = fmaxf(y.y, -126.0f);
    const v2f n = __builtin_elementwise_rint(y);
    const v2f f = y - n;
    v2f p = (v2f)(0x1.41a6fep-13f);
    p = __builtin_elementwise_fma(p, f, (v2f)(0x1.5f44f0p-10f));
    p = __builtin_elementwise_fma(p, f, (v2f)(0x1.3b2dfep-7f));
    p = __builtin_elementwise_fma(p, f, (v2f)(0x1.c6aed6p-5f));
    p = __builtin_elementwise_fma(p, f, (v2f)(0x1.ebfbdap-3f));
    p = __builtin_elementwise_fma(p, f, (v2f)(0x1.62e430p-1f));
    p = __builtin_elementwise_fma(p, f, (v2f)(1.0f));
    return v2f{__builtin_ldexpf(p.x, (int)n.x), __builtin_ldexpf(p.y, (int)n.y)};
}

// 1 / x, correctly rounded, for x in [2^-10, 1] (here x = 1 - alpha in [0.001, 0.9961]): hardware reciprocal (1 ulp), one Newton
// step, one Markstein correction -- all in packed fma.  No scaling or special cases are needed on this range; the oracle divides
// (IEEE), and tests/test_gpu_rasterizer.py::test_rcp_exact_exhaustive checks every float of the range through gs_selftest_rcp.
__device__ __forceinline__ v2f gs_rcp_exact2(v2f x)
{
#pragma clang fp contract(off)
    const v2f r0 = v2f{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
    const v2f one = (v2f)(1.0f);
    const v2f e0 = __builtin_elementwise_fma(-x, r0, one);
    const v2f r1 = __builtin_elementwise_fma(r0, e0, r0);
    const v2f e1 = __builtin_elementwise_fma(-x, r1, one);
    return __builtin_elementwise_fma(e1, r1, r1);
}

// wave-private LDS queue of culled records (ring of 128): appended in stream order, consumed 64 at a time
struct LaneQueue {
    float4* a;      // {mx, my, 0.5a, b}
    float2* b;      // {0.5c, opacity}   (the extents hx, hy of the stream record serve the cull only)
    float4* c;      // {c0, c1, c2, bits(g)}
    int* idx;       // stream index
};
static constexpr int GS_LANES_Q = 128;
static constexpr int GS_LANES_Q_BYTES = GS_LANES_Q * (16 + 16 + 8 + 4);    // per wave

__device__ __forceinline__ LaneQueue lane_queue(unsigned char* base)
{
    LaneQueue q;
    q.a = (float4*)base; q.c = q.a + GS_LANES_Q; q.b = (float2*)(q.c + GS_LANES_Q); q.idx = (int*)(q.b + GS_LANES_Q);
    return q;
}

// Raw batch of the stream held in registers while its loads are in flight.  THREE are kept (r[0..2], consumed round-robin by
// LANES_FILL below): rotating them through "pf0 = pf1; pf1 = pf2" cost 18 v_mov_b64 per raw batch, a quarter of the fill loop.
typedef float v4f __attribute__((ext_vector_type(4)));
struct RawBatch { v4f r0, r1, r2; };
// The three loads are INLINE ASSEMBLY, and so is the wait for them (raw_wait): the compiler's wait-count insertion cannot see
// through the dynamic `phase` of LANES_FILL and waited for (almost) everything in flight before every cull -- s_waitcnt
// vmcnt(1) -- so each raw batch cost a full memory latency (~900 cycles; a third of the time of a silhouette tile).  The count
// that is actually needed is static: whichever of the three batches is consumed, exactly two younger ones (6 loads) are in
// flight, loads return in order, hence vmcnt(6).  Out-of-range lanes load record 0 so that the count holds at the list's end.
// (Stores and atomics also count in vmcnt on gfx9 and may complete out of order with the loads: they can only make the wait
// longer, never shorter than the three oldest loads -- see DESIGN.md section 4.)
__device__ __forceinline__ void raw_load(RawBatch& b, const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                                         const float4* __restrict__ rec2, int idx, bool in_range)
{
    const int safe = in_range ? idx : 0;
    const float4 *p0 = rec0 + safe, *p1 = rec1 + safe, *p2 = rec2 + safe;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(b.r0) : "v"(p0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(b.r1) : "v"(p1) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(b.r2) : "v"(p2) : "memory");
}
// Before a kernel leaves its batch loop: the last reloads are never consumed, so for the compiler their registers die at the load
// and would be handed to other values while the data is still on its way -- which then lands on top of them.  Keep the nine
// registers allocated until everything has arrived (free in practice: the last walk has run in between).
__device__ __forceinline__ void raw_drain(RawBatch& a, RawBatch& b, RawBatch& c)
{
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(a.r0), "+v"(a.r1), "+v"(a.r2), "+v"(b.r0), "+v"(b.r1), "+v"(b.r2), "+v"(c.r0), "+v"(c.r1), "+v"(c.r2) : : "memory");
}
__device__ __forceinline__ void raw_wait(RawBatch& b)
{
#ifndef GS_RAW_WAIT
#define GS_RAW_WAIT "s_waitcnt vmcnt(6)"
#endif
    asm volatile(GS_RAW_WAIT : "+v"(b.r0), "+v"(b.r1), "+v"(b.r2) : : "memory");
}

// centre and half-extent (pixel-centre coordinates) of the bounding rectangle of the active lanes of a quadrant wave
__device__ __forceinline__ void active_rect_c(unsigned long long act, int qx0, int qy0, float& cx, float& cy, float& ex, float& ey)
{
    int xmin, xmax, ymin, ymax;
    active_rect_i(act, xmin, xmax, ymin, ymax);
    cx = (float)qx0 + 0.5f + 0.5f * (float)(xmin + xmax); ex = 0.5f * (float)(xmax - xmin);
    cy = (float)qy0 + 0.5f + 0.5f * (float)(ymin + ymax); ey = 0.5f * (float)(ymax - ymin);
}

// cull one raw batch against the active rectangle and append the survivors to the queue; returns their number (wave-uniform).
// |mx - cx| <= hx + ex is the interval-overlap test (mx + hx >= x0 && mx - hx <= x1) in two instructions per axis; the extents
// carry 0.02 px + 0.05 % of slack (alpha_extent), far above the rounding of either form.
__device__ __forceinline__ int lanes_cull_append(const LaneQueue& q, const RawBatch& cur, bool in_range, int idx, int lane, float cx,
                                                 float cy, float ex, float ey, int qtail)
{
    const float mx = cur.r0.x, my = cur.r0.y, hx = cur.r1.z, hy = cur.r1.w;
    const bool hit = in_range && (hx >= 0.0f) && (fabsf(mx - cx) <= hx + ex) && (fabsf(my - cy) <= hy + ey);
    const unsigned long long hmask = __ballot(hit);
    if (hit) {
        const int slot = (qtail + __popcll(hmask & ((1ull << lane) - 1ull))) & (GS_LANES_Q - 1);
        q.a[slot] = make_float4(cur.r0.x, cur.r0.y, cur.r0.z, cur.r0.w); q.b[slot] = make_float2(cur.r1.x, cur.r1.y);
        q.c[slot] = make_float4(cur.r2.x, cur.r2.y, cur.r2.z, cur.r2.w); q.idx[slot] = idx;
    }
    return __popcll(hmask);
}

// The fill loop of the lanes kernels, unrolled over the three raw batches in flight.  CONT is the loop condition, STEP(B)
// consumes batch B (cull + append + advance) and reloads it with the batch three steps ahead; `phase` (wave-uniform, kept
// across dense batches) says which of the three is next.
#define LANES_FILL(CONT, STEP)                                                                     \
    while (CONT) {                                                                                 \
        switch (phase) {                                                                           \
        case 0: STEP(raw0); phase = 1; if (!(CONT)) break; [[fallthrough]];                        \
        case 1: STEP(raw1); phase = 2; if (!(CONT)) break; [[fallthrough]];                        \
        default: STEP(raw2); phase = 0;                                                            \
        }                                                                                          \
    }

// A lane without a candidate still fetches A slot (index 31 of the batch) and composites it with weight zero: the slot must
// hold FINITE numbers (0 x NaN would poison the accumulators).  Records written by the cull are finite; what a previous kernel
// left in LDS need not be -- so the queue is zeroed once per wave.
__device__ __forceinline__ void lane_queue_clear(const LaneQueue& q, int n_slots, int lane)
{
    for (int i = lane; i < n_slots; i += 64) {
        q.a[i] = make_float4(0.f, 0.f, 0.f, 0.f); q.c[i] = make_float4(0.f, 0.f, 0.f, 0.f); q.b[i] = make_float2(0.f, 0.f); q.idx[i] = 0;
    }
}

__device__ __forceinline__ void lanes_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// ---------------------------------------------------------------------------------------------------
// Forward with a SLIDING WINDOW of two dense batches (the forward of every D).  In the kernel above a pixel whose list of
// the current dense batch is exhausted idles until the longest list of the quadrant is done (a quarter of the candidate slots
// hold a pair).  Here two batches A and B are resident: a lane that has finished A pops from B, and when A is exhausted
// everywhere B becomes A and a new B is built from the next 64 culled records.  The CPU simulation of the schedule (DESIGN.md
// section 4) gives 40 % slot utilisation against 26 %.  Per-pixel order is unchanged (A before B, ascending inside), so the
// image stays bit-identical.  The record ring holds 192 slots (A + B + one raw batch of survivors) = 8 448 B per wave.
// S4 applied inside the compositor (gs_raster_composite_tone* / gs_raster_bwd_tone*): the engine's main stream is the step's
// critical path, and between the two compositor kernels of a view it ran two tiny dependent launches (tone map forward, tone map
// backward) that each waited 20-80 us for a slot on a GPU the other streams keep full (profiles/r03_concurrency_one_step.txt).
// Forward: the epilogue also writes image[p] = tone(render * exposure | alpha).  Backward: the prologue reads the image cotangent
// and forms v_render / v_alpha itself (the arithmetic of tonemap_bwd3_kernel), one atomic per wave for the exposure gradient.
// D == 3 and no background only (what RenderableAttrs.splat passes).
struct ToneFwd { int mode; const float* exposure; float4* image; };
struct ToneBwd { int mode; const float* exposure; const float* render; const float4* v_image; float* v_exposure; };
// Cull log (round 4): what the forward found out about a quadrant's list, kept for the backward.  For every record that entered a
// dense batch of quadrant q of a tile with a non-empty pixel mask the forward appends {stream index, mask} -- mask = the record's
// {alpha >= 1/255} pixel set clipped to the pixels that were still active when the batch was built -- at
//     [4 * offsets[tile] + q * (tile list length) + k],   k = 0 .. count[4 * tile + q) in stream order.
// The backward walks that list from its end: no raw-batch fill, no cull, no ellipse masks, and a pixel never pops a record that
// lies behind its own termination (5 % of the popped candidates) -- see raster_bwd_log_kernel.
struct CullLog { int32_t* idx; unsigned long long* mask; int32_t* count; };

#ifndef GS_WIN_Q_SLOTS
#define GS_WIN_Q_SLOTS 192
#endif
static constexpr int GS_WIN_Q = GS_WIN_Q_SLOTS;
static constexpr int GS_WIN_Q_BYTES = GS_WIN_Q * (16 + 16 + 8 + 4);

__device__ __forceinline__ int win_wrap(int s) { return s >= GS_WIN_Q ? s - GS_WIN_Q : s; }

__device__ __forceinline__ int win_cull_append(const LaneQueue& q, const RawBatch& cur, bool in_range, int idx, int lane, float cx,
                                               float cy, float ex, float ey, int qtail)
{
    const float mx = cur.r0.x, my = cur.r0.y, hx = cur.r1.z, hy = cur.r1.w;
    const bool hit = in_range && (hx >= 0.0f) && (fabsf(mx - cx) <= hx + ex) && (fabsf(my - cy) <= hy + ey);
    const unsigned long long hmask = __ballot(hit);
    if (hit) {
        const int slot = win_wrap(qtail + __popcll(hmask & ((1ull << lane) - 1ull)));
        q.a[slot] = make_float4(cur.r0.x, cur.r0.y, cur.r0.z, cur.r0.w); q.b[slot] = make_float2(cur.r1.x, cur.r1.y);
        q.c[slot] = make_float4(cur.r2.x, cur.r2.y, cur.r2.z, cur.r2.w); q.idx[slot] = idx;
    }
    return __popcll(hmask);
}

template <int CD>
__global__ void __launch_bounds__(256)
raster_fwd_window_kernel(int W, int H, int tile_w, int n_tiles, int D, const int32_t* __restrict__ tile_order,
                         const float4* __restrict__ rec0, const float4* __restrict__ rec1, const float4* __restrict__ rec2,
                         const float* __restrict__ colors, const float* __restrict__ background, GsCount ic,
                         const int32_t* __restrict__ offsets,
                         float* __restrict__ render, float* __restrict__ alphas, int32_t* __restrict__ last_ids, ToneFwd tone, CullLog log,
                         BwdOrder bo)
{
    const int n_isects = (int)gs_count(ic);
    extern __shared__ __align__(16) unsigned char gs_lds_raw[];
    const int tile = tile_order[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int qx0 = tx * GS_TILE + (wave & 1) * 8, qy0 = ty * GS_TILE + (wave >> 1) * 8;
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float px = (float)pxi + 0.5f, py = (float)pyi + 0.5f;
    const v2f pxy = v2f{px, py};
    LaneQueue q;
    {
        unsigned char* base = gs_lds_raw + (size_t)wave * GS_WIN_Q_BYTES;
        q.a = (float4*)base; q.c = q.a + GS_WIN_Q; q.b = (float2*)(q.c + GS_WIN_Q); q.idx = (int*)(q.b + GS_WIN_Q);
    }
    lane_queue_clear(q, GS_WIN_Q, lane);
    // D > 3 (colours outside the record stream; rfstudio/model/geosplat.py:276-295 renders 14 feature channels): the colours of the two
    // resident dense batches are staged in LDS as planes [2][D][64] -- lane j loads the D colours of record j once per (record,
    // quadrant) when its batch is built -- instead of 4 D bytes gathered from global memory per PAIR (round 5: 2.2 -> see DESIGN, D = 14)
    float* colp = CD > 3 ? (float*)(gs_lds_raw + 4 * (size_t)GS_WIN_Q_BYTES) + (size_t)wave * 2 * D * 64 : nullptr;
    int parA = 0;                                                 // colour plane set of batch A (B: the other one)
    const int start = offsets[tile];
    const int end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];

    float T = 1.0f;
    int cur_idx = 0;
    bool done = !inside;
    float pix[CD];
#pragma unroll
    for (int k = 0; k < CD; ++k) pix[k] = 0.0f;

    int qhead = 0, qcount = 0, nA = 0, nB = 0;                   // wave-uniform: ring start = batch A, records in the ring, batch sizes
    unsigned long long listA = 0ull, listB = 0ull;
    int cur_slot = -1;
    int base = start;
    int log_n = 0;                                                // wave-uniform: entries of this quadrant's cull log
    const size_t log_base = 4 * (size_t)start + (size_t)wave * (size_t)(end > start ? end - start : 0);
    RawBatch raw0, raw1, raw2;
    raw_load(raw0, rec0, rec1, rec2, start + lane, start + lane < end);
    raw_load(raw1, rec0, rec1, rec2, start + 64 + lane, start + 64 + lane < end);
    raw_load(raw2, rec0, rec1, rec2, start + 128 + lane, start + 128 + lane < end);
    int phase = 0;
    for (;;) {
        const unsigned long long act = __ballot(!done);
        if (act == 0ull) break;
        // ---- fill: room for one more raw batch of survivors (64) next to A, B and what is already waiting
        { GS_PHASE_BEGIN();
        float rcx, rcy, rex, rey;
        active_rect_c(act, qx0, qy0, rcx, rcy, rex, rey);
#define WIN_FILL_STEP(B)                                                                                               \
        {                                                                                                              \
            GS_STAT(0, 1);                                                                                             \
            raw_wait(B);                                                                                               \
            const int n_hit = win_cull_append(q, B, base + lane < end, base + lane, lane, rcx, rcy, rex, rey, qhead + qcount); \
            raw_load(B, rec0, rec1, rec2, base + 192 + lane, base + 192 + lane < end);                                 \
            qcount += n_hit;                                                                                           \
            base += 64;                                                                                                \
        }
        LANES_FILL(qcount <= GS_WIN_Q - 64 && base < end, WIN_FILL_STEP)
#undef WIN_FILL_STEP
        GS_PHASE_END(0); }
        if (qcount == 0) break;
        lanes_lds_sync();
        long long _pm0 = 0;
#ifdef GS_RASTER_PHASES
        if (blockIdx.x == 0 && threadIdx.x == 0) _pm0 = (long long)__builtin_readcyclecounter();
#endif
        // ---- (re)build the batches that are missing: A, then B behind it
        int xmin, xmax, ymin, ymax;
        active_rect_i(act, xmin, xmax, ymin, ymax);
        if (nA == 0) {
            nA = qcount < 64 ? qcount : 64;
            const int slot = win_wrap(qhead + lane);
            const float4 a = q.a[slot];
            const float2 b = q.b[slot];
            const unsigned long long pm = record_pixel_mask(lane < nA, a.x, a.y, a.z, a.w, b.x, b.y, qx0, qy0, xmin, xmax, ymin, ymax);
            GS_STAT(1, nA);
            if (log.idx) {
                const unsigned long long lm = pm & act;
                const unsigned long long keep = __ballot(lm != 0ull);
                if (lm != 0ull) {
                    const size_t at = log_base + (size_t)(log_n + __popcll(keep & ((1ull << lane) - 1ull)));
                    log.idx[at] = q.idx[slot]; log.mask[at] = lm;
                }
                log_n += __popcll(keep);
            }
            if (CD > 3) {
                const float* cg = colors + (size_t)__float_as_int(q.c[slot].w) * D;
                float* dst = colp + (size_t)parA * D * 64 + lane;
                for (int k = 0; k < D; ++k) dst[k * 64] = lane < nA ? cg[k] : 0.0f;
            }
            listA = gs_bit_transpose64(pm, lane);
            if (done) listA = 0ull;
#ifdef GS_RASTER_STATS
            GS_STAT2(5, 1); GS_STAT2_ALL(4, __popcll(listA));
#endif
        }
        if (nB == 0 && qcount > nA) {
            nB = (qcount - nA) < 64 ? (qcount - nA) : 64;
            const int slot = win_wrap(qhead + nA + lane);
            const float4 a = q.a[slot];
            const float2 b = q.b[slot];
            const unsigned long long pm = record_pixel_mask(lane < nB, a.x, a.y, a.z, a.w, b.x, b.y, qx0, qy0, xmin, xmax, ymin, ymax);
            GS_STAT(1, nB);
            if (log.idx) {
                const unsigned long long lm = pm & act;
                const unsigned long long keep = __ballot(lm != 0ull);
                if (lm != 0ull) {
                    const size_t at = log_base + (size_t)(log_n + __popcll(keep & ((1ull << lane) - 1ull)));
                    log.idx[at] = q.idx[slot]; log.mask[at] = lm;
                }
                log_n += __popcll(keep);
            }
            if (CD > 3) {
                const float* cg = colors + (size_t)__float_as_int(q.c[slot].w) * D;
                float* dst = colp + (size_t)(parA ^ 1) * D * 64 + lane;
                for (int k = 0; k < D; ++k) dst[k * 64] = lane < nB ? cg[k] : 0.0f;
            }
            listB = gs_bit_transpose64(pm, lane);
            if (done) listB = 0ull;
#ifdef GS_RASTER_STATS
            GS_STAT2(5, 1); GS_STAT2_ALL(4, __popcll(listB));
#endif
        }
#ifdef GS_RASTER_PHASES
        if (blockIdx.x == 0 && threadIdx.x == 0) { const long long _t = (long long)__builtin_readcyclecounter(); atomicAdd(&g_raster_stats[1], (unsigned long long)(_t - _pm0)); _pm0 = _t; }
#endif
        // ---- walk until batch A is exhausted in every lane; lanes that are through with A work on B
        if (CD > 3) lanes_lds_sync();                              // (the colour planes written above are read by other lanes)
        const int baseB = qhead + nA;
        if (__ballot(listA != 0ull) != 0ull) do {                 // (rotated by hand: no copies of the loop-carried state per trip)
            GS_STAT(3, 1);
#ifdef GS_RASTER_PHASES
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_raster_stats[3], 1ull);
#endif
            bool has[2]; int slot[2];
            int cofs[2] = { 0, 0 };                                // D > 3: float offset of the candidate's first colour in the planes
            float4 ca[2], cc[2]; float2 cb[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool useA = listA != 0ull;
                unsigned long long cur = useA ? listA : listB;
                has[u] = cur != 0ull;
                const int j = gs_pop_lowest(cur);
                listA = useA ? cur : listA;
                listB = useA ? listB : cur;
                slot[u] = win_wrap((useA ? qhead : baseB) + j);
                if (CD > 3) cofs[u] = (useA ? parA : (parA ^ 1)) * D * 64 + j;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) { ca[u] = q.a[slot[u]]; cb[u] = q.b[slot[u]]; cc[u] = q.c[slot[u]]; }
            v2f alpha; bool ok[2];
            {
#pragma clang fp contract(off)
                // (packed over (x, y) of one record, see the backward walk: same operations in the same order)
                const v2f d0 = v2f{ca[0].x, ca[0].y} - pxy, d1 = v2f{ca[1].x, ca[1].y} - pxy;
                const v2f p0 = v2f{ca[0].z, ca[0].w} * v2f{d0.x, d0.x}, p1 = v2f{ca[1].z, ca[1].w} * v2f{d1.x, d1.x};
                v2f sigma;
                sigma.x = gs_sigma_xy(d0, p0, cb[0].x);
                sigma.y = gs_sigma_xy(d1, p1, cb[1].x);
                alpha = __builtin_elementwise_min(v2f{cb[0].y, cb[1].y} * gs_exp_neg_live2(sigma), (v2f)(0.999f));
                ok[0] = has[0] && sigma.x >= 0.0f && alpha.x >= GS_ALPHA_MIN;
                ok[1] = has[1] && sigma.y >= 0.0f && alpha.y >= GS_ALPHA_MIN;
            }
            bool stopped = false;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float al = u ? alpha.y : alpha.x;
                const float next_T = T * (1.0f - al);
                const bool live = ok[u] && !done;
                const bool stop = live && next_T <= 1e-4f;
                const bool acc = live && !stop;
#ifdef GS_RASTER_STATS
                if (acc) GS_STAT_ALL(2, 1);
#endif
                const float vis = acc ? al * T : 0.0f;
                if (CD <= 3) {
                    pix[0] = fmaf(cc[u].x, vis, pix[0]);
                    if (CD > 1) pix[1] = fmaf(cc[u].y, vis, pix[1]);
                    if (CD > 2) pix[2] = fmaf(cc[u].z, vis, pix[2]);
                } else {
                    // (no branch around the LDS reads: vis == 0 for a lane without an accepted candidate, the planes hold finite numbers)
                    const float* cg = colp + cofs[u];
#pragma unroll
                    for (int k = 0; k < CD; ++k) if (k < D) pix[k] = fmaf(cg[k * 64], vis, pix[k]);
                }
                T = acc ? next_T : T;
                cur_slot = acc ? slot[u] : cur_slot;
                done = done || stop;
                stopped = stopped || stop;
            }
            listA = stopped ? 0ull : listA;
            listB = stopped ? 0ull : listB;
        } while (__ballot(listA != 0ull) != 0ull);
#ifdef GS_RASTER_PHASES
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_raster_stats[2], (unsigned long long)((long long)__builtin_readcyclecounter() - _pm0));
#endif
        // ---- retire A (its slots are recycled by the next fill): resolve the stream index of what was composited last
        if (cur_slot >= 0) { cur_idx = q.idx[cur_slot]; cur_slot = -1; }
        lanes_lds_sync();
        qhead = win_wrap(qhead + nA);
        qcount -= nA;
        listA = listB; nA = nB;
        listB = 0ull; nB = 0;
        parA ^= 1;                                                 // (B's colour planes become A's)
    }
    raw_drain(raw0, raw1, raw2);
    if (log.count && lane == 0) log.count[4 * tile + wave] = log_n;
    if (bo.bcount) {                                                // file the tile for the backward (struct BwdOrder): block barrier only
        __shared__ int s_cnt[4];
        if (lane == 0) s_cnt[wave] = log_n;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int c = max(max(s_cnt[0], s_cnt[1]), max(s_cnt[2], s_cnt[3]));
            int b = 63;                                              // empty: last
            if (c > 0) {
                const int l2 = 31 - __clz(c);
                const int hb = 2 * l2 + ((l2 > 0 && ((c >> (l2 - 1)) & 1)) ? 1 : 0);
                b = 61 - (hb > 61 ? 61 : hb);
            }
            // (a SECOND forward on one prepared workspace -- bench loops, tests -- counts past n_tiles: never stored, and the
            //  backward's `sum == n_tiles` test rejects the over-counted lists and falls back to the forward's tile order)
            const int k = atomicAdd(bo.bcount + b, 1);
            if (k < n_tiles) bo.blist[(size_t)b * n_tiles + k] = tile;
        }
    }

    if (inside) {
        const size_t pid = (size_t)pyi * W + pxi;
        alphas[pid] = 1.0f - T;
        last_ids[pid] = cur_idx;
#pragma unroll
        for (int k = 0; k < CD; ++k)
            if (k < D) render[pid * D + k] = background ? fmaf(T, background[k], pix[k]) : pix[k];
        if (CD == 3 && tone.image) {                              // D == 3, background == nullptr (checked on the host)
#pragma clang fp contract(off)
            const float e = tone.exposure[0], a = 1.0f - T;
            tone.image[pid] = make_float4(tone_fwd(tone.mode, pix[0] * e), tone_fwd(tone.mode, pix[1] * e), tone_fwd(tone.mode, pix[2] * e),
                                          tone.mode == GS_TONE_NONE ? a * e : a);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Backward WITHOUT atomics in the walk (D <= 3 since round 2; every D since round 5: the `ds_add_f64` accumulator kernel it was derived
// from is gone -- `git show 60bad21:geosplatting_amd/csrc/gs_raster.hip`).  Removal experiment on the kernel above (bench workload): 975 us as
// built, 545 us with the nine ds_add_f64 per pair compiled out -- the accumulation, not the arithmetic, was 44 % of it
// (308 M lane-adds per view at ~0.9 LDS cycles each: three to four pixels of a trip add to the same record).  Here the
// walk only produces TWO scalars per popped pair,  s = v_sigma  (0 when the 0.999 cap was active)  and  f = alpha*T,
// and stores them with a plain ds_write_b64 at  pairbuf[base_j + rank of the pixel inside record j's mask]  -- distinct
// addresses, no conflicts.  Then the lanes switch roles: lane j owns record j of the dense batch, walks the set bits
// of ITS pixel mask, and accumulates the moments  sum s {1, dx, dy, dx^2, dx dy, dy^2}  and  sum f v_render  in
// registers (dx = mx - px is a function of the pixel index), from which the nine gradients follow:
//     v_xy = (2 ha Mx + b My, b Mx + 2 hc My),  v_conic = (Mxx/2, Mxy, Myy/2),  v_opacity = -M0 / opacity.
// The sums are exact per-lane fp32 FMAs in pixel order: the per-(quadrant, Gaussian) result is deterministic.
#ifndef GS_PAIR_CAP_N
#define GS_PAIR_CAP_N 448
#endif
static constexpr int GS_PAIR_CAP = GS_PAIR_CAP_N;                       // pair slots per dense batch (records beyond wait for the next one)
template <int CD>
struct Lanes2Lds {
    static constexpr int NV = 6 + CD;
    static constexpr int OFF_MSK = GS_LANES_Q_BYTES;
    static constexpr int OFF_BASE = OFF_MSK + 64 * 8;
    static constexpr int OFF_PAIR = OFF_BASE + 64 * 4;
    static constexpr int PAIR_BYTES = GS_PAIR_CAP * 8 > 64 * NV * 4 ? GS_PAIR_CAP * 8 : 64 * NV * 4;
    static constexpr int WAVE_BYTES = OFF_PAIR + PAIR_BYTES;
};

template <int CD>
__global__ void __launch_bounds__(256)
raster_bwd_lanes2_kernel(int W, int H, int tile_w, int n_tiles, int D, const int32_t* __restrict__ tile_order,
                         const float4* __restrict__ rec0, const float4* __restrict__ rec1, const float4* __restrict__ rec2,
                         const float* __restrict__ colors /* [V, D]: read for D > 3 only (else they travel in the record stream) */,
                         const float* __restrict__ background, GsCount ic, const int32_t* __restrict__ offsets,
                         const float* __restrict__ alphas, const int32_t* __restrict__ last_ids,
                         const float* __restrict__ v_render, const float* __restrict__ v_alphas,
                         float* __restrict__ v_packed, int rec_stride, ToneBwd tone)
{
    const int n_isects = (int)gs_count(ic);
    using LD = Lanes2Lds<CD>;
    constexpr int NV = 6 + CD;
    constexpr int RPI = 64 / NV;
    extern __shared__ __align__(16) unsigned char gs_lds_raw[];
#ifdef GS_EXP_SKIP_HEAVY
    if ((int)blockIdx.x < GS_EXP_SKIP_HEAVY) return;
#endif
#ifdef GS_EXP_ONLY_HEAVY
    if ((int)blockIdx.x >= GS_EXP_ONLY_HEAVY) return;
#endif
#ifdef GS_EXP_PRIO
    if ((int)blockIdx.x < GS_EXP_PRIO) __builtin_amdgcn_s_setprio(3);
#endif
    const int tile = tile_order[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int qx0 = tx * GS_TILE + (wave & 1) * 8, qy0 = ty * GS_TILE + (wave >> 1) * 8;
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float px = (float)pxi + 0.5f, py = (float)pyi + 0.5f;
    const v2f pxy = v2f{px, py};

    const int start = offsets[tile];
    const int end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];
    GS_TL_BEGIN(end - start);
    // (an empty tile has render = 0 and alpha = 0 without a background: its pixels add nothing to the exposure gradient either)
    if (end <= start) { GS_TL_END(); return; }

    // D > 3 (round 5; rfstudio/model/geosplat.py:276-295 renders 14 feature channels): two more plane sets per wave, sized by the
    // run-time D -- colp[D][64], the colours of the dense batch (one load per (record, quadrant) instead of a gather per pair), and
    // vrp[D][64], the pixels' cotangents (the record lanes read them in the reduction: D ds_bpermute per pair otherwise)
    const int wave_bytes = LD::WAVE_BYTES + (CD > 3 ? 2 * D * 64 * 4 : 0);
    unsigned char* wbase = gs_lds_raw + (size_t)wave * wave_bytes;
    const LaneQueue q = lane_queue(wbase);
    lane_queue_clear(q, GS_LANES_Q, lane);
    unsigned long long* msk = (unsigned long long*)(wbase + LD::OFF_MSK);
    int* pbase = (int*)(wbase + LD::OFF_BASE);
    float2* pairbuf = (float2*)(wbase + LD::OFF_PAIR);
    float* stage = (float*)(wbase + LD::OFF_PAIR);              // aliases pairbuf (separated by wave syncs)
    float* colp = (float*)(wbase + LD::WAVE_BYTES);
    float* vrp = colp + (CD > 3 ? D * 64 : 0);

    float T_final = 1.0f, v_a = 0.0f, v_exp = 0.0f;
    int bin_final = -1;
    float v_rc[CD];
#pragma unroll
    for (int k = 0; k < CD; ++k) v_rc[k] = 0.0f;
    if (inside) {
        const size_t pid = (size_t)pyi * W + pxi;
        const float a_out = alphas[pid];
        T_final = 1.0f - a_out;
        bin_final = last_ids[pid];
        if (CD == 3 && tone.v_image) {                            // S4 backward here (tonemap_bwd3_kernel's arithmetic, same order)
#pragma clang fp contract(off)
            const float e = tone.exposure[0];
            const float r = tone.render[3 * pid], gch = tone.render[3 * pid + 1], bl = tone.render[3 * pid + 2];
            const float4 g = tone.v_image[pid];
            const float gx = g.x * tone_grad(tone.mode, r * e), gy = g.y * tone_grad(tone.mode, gch * e), gz = g.z * tone_grad(tone.mode, bl * e);
            v_rc[0] = gx * e; v_rc[1] = gy * e; v_rc[2] = gz * e;
            v_a = tone.mode == GS_TONE_NONE ? g.w * e : g.w;
            v_exp = gx * r + gy * gch + gz * bl + (tone.mode == GS_TONE_NONE ? g.w * a_out : 0.0f);
        } else {
            v_a = v_alphas[pid];
#pragma unroll
            for (int k = 0; k < CD; ++k) if (k < D) v_rc[k] = v_render[pid * D + k];
        }
    }
    if (CD == 3 && tone.v_image) {                                // one exposure-gradient atomic per quadrant wave
        v_exp = gs_wave_sum(v_exp);
        if (lane == 0 && v_exp != 0.0f) gs_atomic_add(tone.v_exposure, v_exp);
    }
    float bg_dot = 0.0f;
    if (background) {
#pragma unroll
        for (int k = 0; k < CD; ++k) if (k < D) bg_dot += background[k] * v_rc[k];
    }
    if (CD > 3) for (int k = 0; k < D; ++k) vrp[k * 64 + lane] = v_rc[k < CD ? k : 0];
    float T = T_final;
    // d(render . v_render + alpha_out v_a)/d(alpha_i) (1 - alpha_i) = T_i (c_i . v) - sum_{j behind i} f_j (c_j . v) + T_final (v_a - bg . v),
    // f = alpha T: the colours enter only through their projection on this pixel's v_render, so ONE accumulator
    // zacc = T_final (v_a - bg . v) - sum f_j (c_j . v) replaces the per-channel buffers of gsplat's formulation (18 -> 6 operations
    // per candidate; same sum, associated differently)
    float zacc = T_final * v_a - T_final * bg_dot;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;

    int top = bin_final;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) top = max(top, __shfl_xor(top, off, 64));
    if (top >= end) top = end - 1;

    int qhead = 0, qcount = 0;                                    // wave-uniform
    // three raw batches in flight, branch-free loads, consumed round-robin (see the forward kernel)
    top = __builtin_amdgcn_readfirstlane(top);
    RawBatch raw0, raw1, raw2;
    raw_load(raw0, rec0, rec1, rec2, top - lane, top - lane >= start);
    raw_load(raw1, rec0, rec1, rec2, top - 64 - lane, top - 64 - lane >= start);
    raw_load(raw2, rec0, rec1, rec2, top - 128 - lane, top - 128 - lane >= start);
    int phase = 0;
    for (;;) {
        // ---- fill: cull raw batches (walking DOWN the list) into the queue
        { GS_PHASE_BEGIN();
        // pixels whose last composited entry lies at or after a batch's lowest index can be valid in it
#define BWD_FILL_STEP(B)                                                                                               \
        {                                                                                                              \
            const unsigned long long act_b = __ballot(bin_final >= top - 63);                                          \
            int n_hit = 0;                                                                                             \
            raw_wait(B);                                                                                               \
            if (act_b != 0ull) {                                                                                       \
                float rcx, rcy, rex, rey;                                                                              \
                active_rect_c(act_b, qx0, qy0, rcx, rcy, rex, rey);                                                    \
                GS_STAT(4, 1);                                                                                         \
                n_hit = lanes_cull_append(q, B, top - lane >= start, top - lane, lane, rcx, rcy, rex, rey, qhead + qcount); \
            }                                                                                                          \
            raw_load(B, rec0, rec1, rec2, top - 192 - lane, top - 192 - lane >= start);                                \
            qcount += n_hit;                                                                                           \
            top -= 64;                                                                                                 \
        }
        LANES_FILL(qcount < 64 && top >= start, BWD_FILL_STEP)
#undef BWD_FILL_STEP
        GS_PHASE_END(0); }
        if (qcount == 0) break;
        int nb = qcount < 64 ? qcount : 64;
        lanes_lds_sync();
        GS_PHASE_BEGIN();
        // ---- dense batch: lane j owns queue slot qhead + j (stream indices DEcrease with j)
        const int idx_low = q.idx[(qhead + nb - 1) & (GS_LANES_Q - 1)];
        const bool live = bin_final >= idx_low;
        const unsigned long long act = __ballot(live);
        const int myslot = (qhead + lane) & (GS_LANES_Q - 1);
        const float4 ra4 = q.a[myslot];                           // the record this lane OWNS in the reduction
        const float2 rb4 = q.b[myslot];
        unsigned long long pm = 0ull;
        if (act != 0ull) {
            int xmin, xmax, ymin, ymax;
            active_rect_i(act, xmin, xmax, ymin, ymax);
            pm = record_pixel_mask(lane < nb, ra4.x, ra4.y, ra4.z, ra4.w, rb4.x, rb4.y, qx0, qy0, xmin, xmax, ymin, ymax) & act;
        }
        // pair slots: exclusive scan of the mask sizes; records whose pairs do not fit wait for the next dense batch
        const int cnt = __popcll(pm);
        int cum = cnt;
        cum = gs_wave_incl_scan(cum);
        if (__ballot(cum > GS_PAIR_CAP) != 0ull) {
            nb = min(nb, __popcll(__ballot(cum <= GS_PAIR_CAP)));   // cum is monotone: a prefix of the records fits (>= 8 of them)
            if (lane >= nb) pm = 0ull;
        }
        msk[lane] = pm;
        pbase[lane] = cum - cnt;
        if (CD > 3) {                                              // the batch's colours (queue slots beyond nb hold earlier, finite records)
            const float* cg = colors + (size_t)__float_as_int(q.c[myslot].w) * D;
            for (int k = 0; k < D; ++k) colp[k * 64 + lane] = lane < nb ? cg[k] : 0.0f;
        }
        unsigned long long list = gs_bit_transpose64(pm, lane);
        GS_STAT(5, nb);
#ifdef GS_RASTER_STATS
        GS_STAT2(1, 1);
        GS_STAT2_ALL(0, __popcll(list));
        { int _mx = __popcll(list);
          for (int _o = 32; _o >= 1; _o >>= 1) _mx = max(_mx, __shfl_xor(_mx, _o, 64));
          GS_STAT2(6, _mx); }
#endif
        lanes_lds_sync();
        GS_PHASE_END(1);
        long long _pw0 = 0;
#ifdef GS_RASTER_PHASES
        if (blockIdx.x == 0 && threadIdx.x == 0) { _pw0 = (long long)__builtin_readcyclecounter(); atomicAdd(&g_raster_stats[6], 1ull); }
#endif
        // Two candidates per lane and trip.  Phase A (fetch, alpha, 1/(1 - alpha)) is independent per candidate and runs in packed
        // fp32; phase B is the serial transmittance recurrence, without branches: a rejected candidate leaves T and the colour
        // buffer unchanged and stores a zero pair.
        while (__ballot(list != 0ull) != 0ull) {
            GS_STAT(6, 1);
#ifdef GS_RASTER_PHASES
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_raster_stats[5], 1ull);
#endif
            const bool has0 = list != 0ull;
            const int j0 = gs_pop_lowest(list);
            const bool has1 = list != 0ull;
            const int j1 = gs_pop_lowest(list);
            const int slot0 = (qhead + j0) & (GS_LANES_Q - 1), slot1 = (qhead + j1) & (GS_LANES_Q - 1);
            const float4 a0 = q.a[slot0], a1 = q.a[slot1];
            const float2 b0 = q.b[slot0], b1 = q.b[slot1];
            const float4 c0 = q.c[slot0], c1 = q.c[slot1];
            const int idx0 = q.idx[slot0], idx1 = q.idx[slot1];
            const int e0 = pbase[j0] + __popcll(msk[j0] & lane_lt), e1 = pbase[j1] + __popcll(msk[j1] & lane_lt);
            v2f sigma, ov, alpha, ra;
            {
#pragma clang fp contract(off)
                // sigma per candidate, packed over (x, y) of ONE record -- the operands arrive that way from the 16-byte LDS read; packing
                // over the two candidates cost nine v_mov per trip to pair their fields up.  Same operations, same order:
                // sigma = fma(ha dx, dx, fma(hc dy, dy, (cb dx) dy)).
                const v2f d0 = v2f{a0.x, a0.y} - pxy, d1 = v2f{a1.x, a1.y} - pxy;                 // {dx, dy}
                const v2f p0 = v2f{a0.z, a0.w} * v2f{d0.x, d0.x}, p1 = v2f{a1.z, a1.w} * v2f{d1.x, d1.x};   // {ha dx, cb dx}
                sigma.x = gs_sigma_xy(d0, p0, b0.x);
                sigma.y = gs_sigma_xy(d1, p1, b1.x);
                ov = v2f{b0.y, b1.y} * gs_exp_neg_live2(sigma);
                alpha = __builtin_elementwise_min(ov, (v2f)(0.999f));
                ra = gs_rcp_exact2((v2f)(1.0f) - alpha);
            }
            const bool ok0 = has0 && idx0 <= bin_final && sigma.x >= 0.0f && alpha.x >= GS_ALPHA_MIN;
            const bool ok1 = has1 && idx1 <= bin_final && sigma.y >= 0.0f && alpha.y >= GS_ALPHA_MIN;
            {
                const float Tn = T * ra.x;
                const float fac = ok0 ? alpha.x * Tn : 0.0f;
                float cv;
                if (CD <= 3) {
                    cv = c0.x * v_rc[0];
                    if (CD > 1) cv = fmaf(c0.y, v_rc[1], cv);
                    if (CD > 2) cv = fmaf(c0.z, v_rc[2], cv);
                } else {
                    cv = 0.0f;
#pragma unroll
                    for (int k = 0; k < CD; ++k) if (k < D) cv = fmaf(colp[k * 64 + j0], v_rc[k], cv);
                }
                const float v_alpha = fmaf(Tn, cv, ra.x * zacc);
                const float s_out = (ok0 && ov.x <= 0.999f) ? -ov.x * v_alpha : 0.0f;
                zacc = fmaf(-fac, cv, zacc);
                T = ok0 ? Tn : T;
#ifdef GS_RASTER_STATS
                if (ok0) GS_STAT_ALL(7, 1);
                if (has0 && !ok0) { if (idx0 > bin_final) GS_STAT2_ALL(2, 1); else GS_STAT2_ALL(3, 1); }
#endif
                if (has0) pairbuf[e0] = make_float2(s_out, fac);
            }
            {
                const float Tn = T * ra.y;
                const float fac = ok1 ? alpha.y * Tn : 0.0f;
                float cv;
                if (CD <= 3) {
                    cv = c1.x * v_rc[0];
                    if (CD > 1) cv = fmaf(c1.y, v_rc[1], cv);
                    if (CD > 2) cv = fmaf(c1.z, v_rc[2], cv);
                } else {
                    cv = 0.0f;
#pragma unroll
                    for (int k = 0; k < CD; ++k) if (k < D) cv = fmaf(colp[k * 64 + j1], v_rc[k], cv);
                }
                const float v_alpha = fmaf(Tn, cv, ra.y * zacc);
                const float s_out = (ok1 && ov.y <= 0.999f) ? -ov.y * v_alpha : 0.0f;
                zacc = fmaf(-fac, cv, zacc);
                T = ok1 ? Tn : T;
#ifdef GS_RASTER_STATS
                if (ok1) GS_STAT_ALL(7, 1);
                if (has1 && !ok1) { if (idx1 > bin_final) GS_STAT2_ALL(2, 1); else GS_STAT2_ALL(3, 1); }
#endif
                if (has1) pairbuf[e1] = make_float2(s_out, fac);
            }
        }
        lanes_lds_sync();
#ifdef GS_RASTER_PHASES
        if (blockIdx.x == 0 && threadIdx.x == 0) { const long long _t = (long long)__builtin_readcyclecounter(); atomicAdd(&g_raster_stats[2], (unsigned long long)(_t - _pw0)); _pw0 = _t; }
#endif
        // ---- reduction: lane j sums the pairs of record j over the set bits of its pixel mask (pixel order)
        // (moments in packed fp32: {1, dx, dy} and {dx dx, dx dy, dy dy} weighted by s, the colour gradients weighted by f)
        float sum[NV];
        {
            const float X = ra4.x - ((float)qx0 + 0.5f), Y = ra4.y - ((float)qy0 + 0.5f);    // dx = X - x,  dy = Y - y
            unsigned long long m = pm;
            int e = cum - cnt;
            float m0 = 0.0f, mxy = 0.0f, c2 = 0.0f;
            v2f m1 = (v2f)(0.0f), m2 = (v2f)(0.0f), c01 = (v2f)(0.0f);
            float csum[CD > 3 ? CD : 1];
#pragma unroll
            for (int k = 0; k < (CD > 3 ? CD : 1); ++k) csum[k] = 0.0f;
            while (__ballot(m != 0ull) != 0ull) {
                GS_STAT2(7, 1);
#ifdef GS_RASTER_PHASES
                if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_raster_stats[7], 1ull);
#endif
                const bool has = m != 0ull;
                const int p = gs_pop_lowest(m);
                const float2 sfr = pairbuf[has ? e : 0];
                float4 vr = make_float4(0.f, 0.f, 0.f, 0.f);           // v_render of pixel p: from that lane's registers (ds_bpermute, no LDS storage)
                if (CD <= 3) { vr.x = __shfl(v_rc[0], p, 64); vr.y = CD > 1 ? __shfl(v_rc[1], p, 64) : 0.0f; vr.z = CD > 2 ? __shfl(v_rc[2], p, 64) : 0.0f; }
                if (CD > 3) {                                          // D > 3: from the cotangent planes
                    const float fw = has ? sfr.y : 0.0f;
#pragma unroll
                    for (int k = 0; k < CD; ++k) if (k < D) csum[k] = fmaf(fw, vrp[k * 64 + p], csum[k]);
                }
                e += has ? 1 : 0;
                const float s_w = has ? sfr.x : 0.0f, f_w = has ? sfr.y : 0.0f;
                const v2f d = v2f{X, Y} - v2f{(float)(p & 7), (float)(p >> 3)};
                const v2f sd = d * s_w;
                m0 += s_w;
                m1 += sd;
                m2 = __builtin_elementwise_fma(sd, d, m2);
                mxy = fmaf(sd.x, d.y, mxy);
                c01 = __builtin_elementwise_fma((v2f)(f_w), v2f{vr.x, vr.y}, c01);
                if (CD > 2) c2 = fmaf(f_w, vr.z, c2);
            }
            sum[0] = m0; sum[1] = m1.x; sum[2] = m1.y; sum[3] = m2.x; sum[4] = mxy; sum[5] = m2.y;
            if (CD <= 3) {
                sum[6] = c01.x;
                if (CD > 1) sum[7] = c01.y;
                if (CD > 2) sum[8] = c2;
            } else {
#pragma unroll
                for (int k = 0; k < CD; ++k) sum[6 + k] = csum[k];
            }
        }
        lanes_lds_sync();                                          // pairbuf is dead: its space becomes the commit staging
#ifdef GS_RASTER_PHASES
        if (blockIdx.x == 0 && threadIdx.x == 0) { const long long _t = (long long)__builtin_readcyclecounter(); atomicAdd(&g_raster_stats[3], (unsigned long long)(_t - _pw0)); _pw0 = _t; }
#endif
        {
            const float ga = ra4.z, gb = ra4.w, gc = rb4.x, go = rb4.y;
            const float M0 = sum[0], Mx = sum[1], My = sum[2];
            float out[NV];
            out[0] = (2.0f * ga) * Mx + gb * My;
            out[1] = gb * Mx + (2.0f * gc) * My;
            out[2] = 0.5f * sum[3]; out[3] = sum[4]; out[4] = 0.5f * sum[5];
            out[5] = (M0 != 0.0f) ? -M0 / go : 0.0f;             // sum of vis * v_alpha over the uncapped pairs
#pragma unroll
            for (int k = 0; k < CD; ++k) out[6 + k] = sum[6 + k];
            if (lane < nb) {
#pragma unroll
                for (int k = 0; k < NV; ++k) stage[lane * NV + k] = out[k];
            }
        }
        lanes_lds_sync();
        // ---- commit: RPI records x NV values per atomic instruction; the NV lanes of a record hit one packed gradient record
        {
            const int r = lane / NV, k = lane - r * NV;
            for (int it = 0; it * RPI < nb; ++it) {
                const int j = it * RPI + r;
                if (r < RPI && j < nb && k < 6 + D) {
                    const float v = stage[j * NV + k];
                    if (v != 0.0f) {
                        const int g = __float_as_int(q.c[(qhead + j) & (GS_LANES_Q - 1)].w);
                        gs_atomic_add(v_packed + (size_t)g * rec_stride + k, v);
                    }
                }
            }
        }
        lanes_lds_sync();
#ifdef GS_RASTER_PHASES
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_raster_stats[4], (unsigned long long)((long long)__builtin_readcyclecounter() - _pw0));
#endif
        qhead = (qhead + nb) & (GS_LANES_Q - 1);
        qcount -= nb;
    }
    raw_drain(raw0, raw1, raw2);
    GS_TL_END();
}

// ---------------------------------------------------------------------------------------------------
// Backward from the forward's CULL LOG (round 4; D <= 3).  raster_bwd_lanes2_kernel repeats, walking down the tile's list, what the
// forward already did walking up: cull every raw batch against the quadrant, compact the survivors into a queue, solve every
// survivor's ellipse for its pixel mask -- a fifth of its instructions -- and then pops 5 % of its candidates only to find that the
// pixel had terminated in front of them.  Here the forward leaves {stream index, pixel mask} of every record that entered one of
// its dense batches (struct CullLog), and this kernel takes 64 log entries at a time from the END of the quadrant's log:
//   * no fill loop, no cull, no ring queue: lane j gathers "its" record by stream index (two batches ahead: entries, one batch
//     ahead: the three 16-byte record parts) and writes queue slot j;
//   * no record_pixel_mask: the logged mask, clipped to the pixels whose last composited entry lies at or behind the batch;
//   * the queue needs 64 slots instead of 128, which buys 798 (+ 2 spare) pair slots instead of 448 in the same 9 984 bytes of LDS per wave:
//     a dense batch is 64 records almost always (lanes2: 47 on average, i.e. 36 % more batches with their fixed costs).
// Walk, record-lane reduction and commit are those of lanes2 (same arithmetic, same per-pixel order).
// Round 6, built, correct (every gradient test green), measured and removed (`git show 35b5b24:geosplatting_amd/csrc/gs_raster.hip`,
// raster_bwd_flat_kernel): the walk FLATTENED -- the pairs of a sub-batch laid out pixel after pixel as slots of two consecutive
// candidates, every trip working 64 slots of whatever pixels, the transmittance / accumulator recurrences as two segmented scans over
// the wave (DPP inside the rows, read-lanes across them, scalar flag masks), pixel state carried between chunks.  80.5 % of its candidate
// slots hold a pair (per-pixel walk: 35.8 %) and it takes 333 k trips of 179 VALU instructions instead of 747 k of ~115 -- but the slot
// codes cost an expansion loop per sub-batch, the codes' LDS leaves 616 instead of 790 pair slots (more sub-batches, each with its
// transposes, scans, reduction set-up and commit), and the kernel ends at 169 M VALU wave instructions against 152 M: 0.43 ms alone
// (0.45 with the chunks decoupled through carried state and late pulls) against 0.385, 671-684 against 698-702 views/s.  LDS bank
// conflicts were NOT the reason (20.0 M against 20.8 M conflict cycles, rocprofv3 --pmc).
#ifndef GS_LOG_PAIR_SLOTS
#define GS_LOG_PAIR_SLOTS 790
#endif
static constexpr int GS_LOG_PAIR_CAP = GS_LOG_PAIR_SLOTS;                       // + two spare slots (stores of lanes without a candidate land in [790])
#ifndef GS_LOG_ASSIST_MAX                                         //   + 64 bytes: the pixels of an assisted walk trip
#define GS_LOG_ASSIST_MAX 16                                      // walk trips with at most this many busy pixels are ASSISTED (0: never)
#endif
struct LogLds {
    static constexpr int Q_BYTES = 64 * (16 + 16 + 8 + 4);
    static constexpr int OFF_MSK = Q_BYTES;
    static constexpr int OFF_BASE = OFF_MSK + 64 * 8;
    static constexpr int OFF_PAIR = OFF_BASE + 64 * 4;
    static constexpr int OFF_APIX = OFF_PAIR + (GS_LOG_PAIR_CAP + 2) * 8;
    static constexpr int WAVE_BYTES = OFF_APIX + 16 * 4;
};

// DPP shift inside a 16-lane row: lane l receives src of lane l - N; lanes whose source lies outside the row receive `ident`
template <int CTRL>
__device__ __forceinline__ float gs_row_shr(float ident, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
}

template <int CD>
__global__ void __launch_bounds__(256)
raster_bwd_log_kernel(int W, int H, int tile_w, int n_tiles, int D, const int32_t* __restrict__ tile_order,
                      const float4* __restrict__ rec0, const float4* __restrict__ rec1, const float4* __restrict__ rec2,
                      const float* __restrict__ background, GsCount ic, const int32_t* __restrict__ offsets,
                      const float* __restrict__ alphas, const int32_t* __restrict__ last_ids,
                      const float* __restrict__ v_render, const float* __restrict__ v_alphas,
                      float* __restrict__ v_packed, int rec_stride, ToneBwd tone, CullLog log, BwdOrder bo)
{
    const int n_isects = (int)gs_count(ic);
#if defined(GS_RASTER_STATS) && !defined(GS_RASTER_PHASES)
    int walk_hist[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };                // wave-uniform; flushed once at the end (bank 3)
#endif
    static_assert(CD <= 3, "colours come from the record stream (D <= 3)");
    using LD = LogLds;
    constexpr int NV = 6 + CD;
    constexpr int RPI = 64 / NV;
    static_assert(64 * NV * 4 <= GS_LOG_PAIR_CAP * 8, "the commit staging aliases the pair buffer");
    extern __shared__ __align__(16) unsigned char gs_lds_raw[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int tile;
    if (bo.bcount) {                                               // block b = the b-th tile of the bucket lists, longest bucket first
        const int c = bo.bcount[lane];
        int incl = c;
        incl = gs_wave_incl_scan(incl);
        const unsigned long long above = __ballot(incl > (int)blockIdx.x);
        const int b = above ? __builtin_ctzll(above) : 63;
        const int before = __builtin_amdgcn_readlane(incl - c, b);
        // (the lists hold every tile once when ONE forward ran behind gs_raster_prepare*; anything else: the forward's own order)
        tile = __builtin_amdgcn_readlane(incl, 63) == n_tiles
                   ? __builtin_amdgcn_readfirstlane(bo.blist[(size_t)b * n_tiles + ((int)blockIdx.x - before)]) : tile_order[blockIdx.x];
    } else {
        tile = tile_order[blockIdx.x];
    }
    const int tx = tile % tile_w, ty = tile / tile_w;
    const int qx0 = tx * GS_TILE + (wave & 1) * 8, qy0 = ty * GS_TILE + (wave >> 1) * 8;
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    const float px = (float)pxi + 0.5f, py = (float)pyi + 0.5f;
    const v2f pxy = v2f{px, py};

    const int start = offsets[tile];
    const int end = (tile == n_tiles - 1) ? n_isects : offsets[tile + 1];
    GS_TL_BEGIN(end - start);
    if (end <= start) { GS_TL_END(); return; }

    unsigned char* wbase = gs_lds_raw + (size_t)wave * LD::WAVE_BYTES;
    float4* qa = (float4*)wbase; float4* qc = qa + 64; float2* qb = (float2*)(qc + 64); int* qidx = (int*)(qb + 64);
    unsigned long long* msk = (unsigned long long*)(wbase + LD::OFF_MSK);
    int* pbase = (int*)(wbase + LD::OFF_BASE);
    float2* pairbuf = (float2*)(wbase + LD::OFF_PAIR);
    int* apix = (int*)(wbase + LD::OFF_APIX);
    float* stage = (float*)(wbase + LD::OFF_PAIR);              // aliases pairbuf (separated by wave syncs)

    float T_final = 1.0f, v_a = 0.0f, v_exp = 0.0f;
    int bin_final = -1;
    float v_rc[CD];
#pragma unroll
    for (int k = 0; k < CD; ++k) v_rc[k] = 0.0f;
    if (inside) {
        const size_t pid = (size_t)pyi * W + pxi;
        const float a_out = alphas[pid];
        T_final = 1.0f - a_out;
        bin_final = last_ids[pid];
        if (CD == 3 && tone.v_image) {                            // S4 backward here (tonemap_bwd3_kernel's arithmetic, same order)
#pragma clang fp contract(off)
            const float e = tone.exposure[0];
            const float r = tone.render[3 * pid], gch = tone.render[3 * pid + 1], bl = tone.render[3 * pid + 2];
            const float4 g = tone.v_image[pid];
            const float gx = g.x * tone_grad(tone.mode, r * e), gy = g.y * tone_grad(tone.mode, gch * e), gz = g.z * tone_grad(tone.mode, bl * e);
            v_rc[0] = gx * e; v_rc[1] = gy * e; v_rc[2] = gz * e;
            v_a = tone.mode == GS_TONE_NONE ? g.w * e : g.w;
            v_exp = gx * r + gy * gch + gz * bl + (tone.mode == GS_TONE_NONE ? g.w * a_out : 0.0f);
        } else {
            v_a = v_alphas[pid];
#pragma unroll
            for (int k = 0; k < CD; ++k) if (k < D) v_rc[k] = v_render[pid * D + k];
        }
    }
    if (CD == 3 && tone.v_image) {                                // one exposure-gradient atomic per quadrant wave
        v_exp = gs_wave_sum(v_exp);
        if (lane == 0 && v_exp != 0.0f) gs_atomic_add(tone.v_exposure, v_exp);
    }
    float bg_dot = 0.0f;
    if (background) {
#pragma unroll
        for (int k = 0; k < CD; ++k) if (k < D) bg_dot += background[k] * v_rc[k];
    }
    float T = T_final;
    float zacc = T_final * v_a - T_final * bg_dot;              // see raster_bwd_lanes2_kernel
    const unsigned long long lane_lt = (1ull << lane) - 1ull;

    int pos = __builtin_amdgcn_readfirstlane(log.count[4 * tile + wave]);       // log entries not yet taken (wave-uniform)
    if (pos <= 0) { GS_TL_END(); return; }
    const size_t log_base = 4 * (size_t)start + (size_t)wave * (size_t)(end - start);
    const int32_t* lidx = log.idx + log_base;
    const unsigned long long* lmsk = log.mask + log_base;
    // software pipeline: entries two batches ahead, records one batch ahead.  Lanes past the log's start re-read entry 0 (a valid
    // record: finite numbers for the zero-weight slots) and are masked out.
    int e_idx, g_idx; unsigned long long e_msk, g_msk;
    float4 g0, g1, g2;
    {
        const int e = pos - 1 - lane;
        const int es = e >= 0 ? e : 0;
        g_idx = lidx[es]; g_msk = e >= 0 ? lmsk[es] : 0ull;
        g0 = rec0[g_idx]; g1 = rec1[g_idx]; g2 = rec2[g_idx];
        const int e2 = pos - 65 - lane;
        const int es2 = e2 >= 0 ? e2 : 0;
        e_idx = lidx[es2]; e_msk = e2 >= 0 ? lmsk[es2] : 0ull;
    }
    while (pos > 0) {
        const int nb = pos < 64 ? pos : 64;
        // ---- issue the loads of the batches behind this one, then publish this batch's records in the queue
        const float4 c0 = g0, c1 = g1, c2 = g2;
        const int c_idx = g_idx;
        const unsigned long long c_msk = g_msk;
        g_idx = e_idx; g_msk = e_msk;
        g0 = rec0[g_idx]; g1 = rec1[g_idx]; g2 = rec2[g_idx];
        {
            const int e2 = pos - 129 - lane;
            const int es2 = e2 >= 0 ? e2 : 0;
            e_idx = lidx[es2]; e_msk = e2 >= 0 ? lmsk[es2] : 0ull;
        }
        pos -= nb;
        GS_PHASE_BEGIN();
        lanes_lds_sync();                                           // (the previous batch's commit has read its queue slots)
        qa[lane] = c0; qb[lane] = make_float2(c1.x, c1.y); qc[lane] = c2; qidx[lane] = c_idx;
        GS_STAT(5, nb);
        const int idx_low = __builtin_amdgcn_readlane(c_idx, nb - 1);
        const bool live = bin_final >= idx_low;
        const unsigned long long act = __ballot(live);
        const unsigned long long pm = lane < nb ? (c_msk & act) : 0ull;
        if (__ballot(pm != 0ull) == 0ull) { GS_PHASE_END(1); continue; }
        const float4 ra4 = c0;                                     // the record this lane OWNS in the reduction
        const float2 rb4 = make_float2(c1.x, c1.y);
        const int cnt = __popcll(pm);
        int cum = cnt;
        cum = gs_wave_incl_scan(cum);
        GS_PHASE_END(1);
        // ---- sub-batches: as many records as the pair buffer holds (almost always all 64)
        int r0 = 0, cumbase = 0;
        while (r0 < nb) {
            const int r1 = r0 + __popcll(__ballot(lane >= r0 && lane < nb && (cum - cumbase) <= GS_LOG_PAIR_CAP));
            const bool mine = lane >= r0 && lane < r1;
            const unsigned long long pms = mine ? pm : 0ull;
            lanes_lds_sync();
            msk[lane] = pms;
            pbase[lane] = cum - cnt - cumbase;
            unsigned long long list = gs_bit_transpose64(pms, lane);
#ifdef GS_RASTER_STATS
            GS_STAT2(1, 1);
            GS_STAT2_ALL(0, __popcll(list));
#endif
            lanes_lds_sync();
            long long _pw0 = 0;
#ifdef GS_RASTER_PHASES
            if (blockIdx.x == 0 && threadIdx.x == 0) { _pw0 = (long long)__builtin_readcyclecounter(); atomicAdd(&g_raster_stats[6], 1ull); }
#endif
            if (__ballot(list != 0ull) != 0ull) do {
                GS_STAT(6, 1);
#if defined(GS_RASTER_STATS) && !defined(GS_RASTER_PHASES) && !defined(GS_HIST_REDUCTION)
                { const int na = __popcll(__ballot(list != 0ull));
#pragma unroll
                  for (int k = 0; k < 8; ++k) walk_hist[k] += (na > 8 * k && na <= 8 * k + 8) ? 1 : 0; }
#endif
#ifdef GS_RASTER_PHASES
                if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_raster_stats[5], 1ull);
#endif
#if GS_LOG_ASSIST_MAX > 0
                // ---- ASSISTED trip.  Half of the walk's trips have at most 16 busy pixels (a third at most 8): the recurrence of
                // a pixel is sequential, but everything in front of it -- sigma, exp, reciprocal, colour dot, the pair slot: 4/5 of
                // a candidate's instructions -- is not.  G = 4 (8) lanes serve one busy pixel: helper h takes the pixel's h-th next
                // candidate, the transmittance / accumulator recurrences become two exclusive scans over the G lanes (DPP inside the
                // row), every helper stores its own pair, and the pixel's lane takes over the state behind the G candidates.
                // Same candidates in the same order as the plain trips; the products are associated differently (a gradient).
                {
                    const unsigned long long am = __ballot(list != 0ull);
                    const int na = __popcll(am);
                    // (lists of one or two candidates: a plain trip finishes them for less)
                    if (na <= GS_LOG_ASSIST_MAX && __ballot(__popcll(list) > 2) != 0ull) {
                        const int gs = na <= 8 ? 3 : 2, G = 1 << gs;
                        const int rank = __popcll(am & lane_lt);
                        lanes_lds_sync();
                        if (list != 0ull) apix[rank] = lane;
                        lanes_lds_sync();
                        const int g = lane >> gs, h = lane & (G - 1);
                        const bool grp = g < na;
                        const int p = grp ? apix[g] : 0;                                   // the pixel (lane) this lane assists
                        const int pa = p << 2;
                        unsigned long long L = ((unsigned long long)(unsigned)__builtin_amdgcn_ds_bpermute(pa, (int)(unsigned)(list >> 32)) << 32) |
                                               (unsigned long long)(unsigned)__builtin_amdgcn_ds_bpermute(pa, (int)(unsigned)list);
                        const float T0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pa, __builtin_bit_cast(int, T)));
                        const float Z0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pa, __builtin_bit_cast(int, zacc)));
                        const int bf = __builtin_amdgcn_ds_bpermute(pa, bin_final);
                        float vr[CD];
#pragma unroll
                        for (int c = 0; c < CD; ++c) vr[c] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pa, __builtin_bit_cast(int, v_rc[c])));
                        if (!grp) L = 0ull;
                        for (int i = 0; i < G - 1; ++i) L = (i < h) ? (L & (L - 1ull)) : L;   // skip the candidates of the helpers in front
                        const bool has = L != 0ull;
                        const int j = gs_pop_lowest(L);                                    // (L is now what remains behind this helper)
                        const float4 a = qa[j]; const float2 b = qb[j]; const float4 cc = qc[j];
                        const int idx = qidx[j];
                        const unsigned long long below = (1ull << p) - 1ull;
                        const int e = pbase[j] + __popcll(msk[j] & below);
                        float sigma, ov, alpha, ra;
                        {
#pragma clang fp contract(off)
                            const v2f d = v2f{a.x, a.y} - v2f{(float)(qx0 + (p & 7)) + 0.5f, (float)(qy0 + (p >> 3)) + 0.5f};
                            const v2f pp = v2f{a.z, a.w} * v2f{d.x, d.x};
                            sigma = gs_sigma_xy(d, pp, b.x);
                            ov = b.y * __builtin_amdgcn_exp2f(sigma * -1.4426950408889634f);
                            alpha = fminf(ov, 0.999f);
                            const float x = 1.0f - alpha;
                            const float r0 = __builtin_amdgcn_rcpf(x);
                            ra = fmaf(r0, fmaf(-x, r0, 1.0f), r0);
                        }
                        const bool ok = has & (idx <= bf) & (sigma >= 0.0f) & (alpha >= GS_ALPHA_MIN);
                        float cv = cc.x * vr[0];
                        if (CD > 1) cv = fmaf(cc.y, vr[1], cv);
                        if (CD > 2) cv = fmaf(cc.z, vr[2], cv);
                        // exclusive product of the multipliers of the helpers in front (inclusive scan, then one more shift)
                        float pi = ok ? ra : 1.0f;
                        { const float t = gs_row_shr<0x111>(1.0f, pi); pi *= (h >= 1) ? t : 1.0f; }
                        { const float t = gs_row_shr<0x112>(1.0f, pi); pi *= (h >= 2) ? t : 1.0f; }
                        { const float t = gs_row_shr<0x114>(1.0f, pi); pi *= (h >= 4) ? t : 1.0f; }
                        float pe = gs_row_shr<0x111>(1.0f, pi); pe = (h >= 1) ? pe : 1.0f;
                        const float Tb = T0 * pe, Tn = Tb * ra;
                        const float fac = ok ? alpha * Tn : 0.0f;
                        float si = fac * cv;
                        { const float t = gs_row_shr<0x111>(0.0f, si); si += (h >= 1) ? t : 0.0f; }
                        { const float t = gs_row_shr<0x112>(0.0f, si); si += (h >= 2) ? t : 0.0f; }
                        { const float t = gs_row_shr<0x114>(0.0f, si); si += (h >= 4) ? t : 0.0f; }
                        float se = gs_row_shr<0x111>(0.0f, si); se = (h >= 1) ? se : 0.0f;
                        const float zb = Z0 - se;
                        const float v_alpha = fmaf(Tn, cv, ra * zb);
                        const float s_out = (ok && ov <= 0.999f) ? -ov * v_alpha : 0.0f;
#ifdef GS_RASTER_STATS
                        if (ok) GS_STAT_ALL(7, 1);
                        if (has && !ok) { if (idx > bf) GS_STAT2_ALL(2, 1); else GS_STAT2_ALL(3, 1); }
#endif
                        pairbuf[has ? e : GS_LOG_PAIR_CAP] = make_float2(s_out, fac);
                        // the pixel's lane takes over the state behind its last helper
                        const float T_end = T0 * pi, Z_end = Z0 - si;
                        const int src = (((rank << gs) + G - 1) & 63) << 2;
                        const float Tg = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, T_end)));
                        const float Zg = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, Z_end)));
                        const unsigned Llo = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)L);
                        const unsigned Lhi = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(L >> 32));
                        if (list != 0ull) { T = Tg; zacc = Zg; list = ((unsigned long long)Lhi << 32) | Llo; }
                        continue;
                    }
                }
#endif
                const bool has0 = list != 0ull;
                const int j0 = gs_pop_lowest(list);
                const bool has1 = list != 0ull;
                const int j1 = gs_pop_lowest(list);
                const float4 a0 = qa[j0], a1 = qa[j1];
                const float2 b0 = qb[j0], b1 = qb[j1];
                const float4 cc0 = qc[j0], cc1 = qc[j1];
                const int idx0 = qidx[j0], idx1 = qidx[j1];
                const int e0 = pbase[j0] + __popcll(msk[j0] & lane_lt), e1 = pbase[j1] + __popcll(msk[j1] & lane_lt);
                v2f sigma, ov, alpha, ra;
                {
#pragma clang fp contract(off)
                    const v2f d0 = v2f{a0.x, a0.y} - pxy, d1 = v2f{a1.x, a1.y} - pxy;                 // {dx, dy}
                    const v2f p0 = v2f{a0.z, a0.w} * v2f{d0.x, d0.x}, p1 = v2f{a1.z, a1.w} * v2f{d1.x, d1.x};   // {ha dx, cb dx}
                    sigma.x = gs_sigma_xy(d0, p0, b0.x);
                    sigma.y = gs_sigma_xy(d1, p1, b1.x);
#ifndef GS_BWD_LOG_EXACT_MATH
                    // hardware exp2 (1 ulp) and a reciprocal with ONE Newton step here, not the canonical exp / correctly rounded
                    // quotient of the forward and of raster_bwd_lanes2_kernel: this is a gradient (1e-4, and upstream's backward
                    // uses __expf itself); the walk loses 13 of its 130 issue slots: 0.50 -> 0.465 ms alone, 592 -> 604 views/s.
                    // Once or twice per view a pair at alpha = 1/255 falls on the other side of the test than in the forward: the
                    // transmittance of ONE pixel is then off by 0.4 % in front of that pair.
                    ov = v2f{b0.y, b1.y} * v2f{__builtin_amdgcn_exp2f(sigma.x * -1.4426950408889634f), __builtin_amdgcn_exp2f(sigma.y * -1.4426950408889634f)};
                    alpha = __builtin_elementwise_min(ov, (v2f)(0.999f));
                    {
                        const v2f x = (v2f)(1.0f) - alpha;
                        const v2f r0 = v2f{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
                        ra = __builtin_elementwise_fma(r0, __builtin_elementwise_fma(-x, r0, (v2f)(1.0f)), r0);
                    }
#else
                    ov = v2f{b0.y, b1.y} * gs_exp_neg_live2(sigma);
                    alpha = __builtin_elementwise_min(ov, (v2f)(0.999f));
                    ra = gs_rcp_exact2((v2f)(1.0f) - alpha);
#endif
                }
                // (bitwise: no short-circuit branches in the walk)
                const bool ok0 = has0 & (idx0 <= bin_final) & (sigma.x >= 0.0f) & (alpha.x >= GS_ALPHA_MIN);
                const bool ok1 = has1 & (idx1 <= bin_final) & (sigma.y >= 0.0f) & (alpha.y >= GS_ALPHA_MIN);
                {
                    const float Tn = T * ra.x;
                    const float fac = ok0 ? alpha.x * Tn : 0.0f;
                    float cv = cc0.x * v_rc[0];
                    if (CD > 1) cv = fmaf(cc0.y, v_rc[1], cv);
                    if (CD > 2) cv = fmaf(cc0.z, v_rc[2], cv);
                    const float v_alpha = fmaf(Tn, cv, ra.x * zacc);
                    const float s_out = (ok0 && ov.x <= 0.999f) ? -ov.x * v_alpha : 0.0f;
                    zacc = fmaf(-fac, cv, zacc);
                    T = ok0 ? Tn : T;
#ifdef GS_RASTER_STATS
                    if (ok0) GS_STAT_ALL(7, 1);
                    if (has0 && !ok0) { if (idx0 > bin_final) GS_STAT2_ALL(2, 1); else GS_STAT2_ALL(3, 1); }
#endif
                    pairbuf[has0 ? e0 : GS_LOG_PAIR_CAP] = make_float2(s_out, fac);        // (a lane without a candidate writes the spare slot)
                }
                {
                    const float Tn = T * ra.y;
                    const float fac = ok1 ? alpha.y * Tn : 0.0f;
                    float cv = cc1.x * v_rc[0];
                    if (CD > 1) cv = fmaf(cc1.y, v_rc[1], cv);
                    if (CD > 2) cv = fmaf(cc1.z, v_rc[2], cv);
                    const float v_alpha = fmaf(Tn, cv, ra.y * zacc);
                    const float s_out = (ok1 && ov.y <= 0.999f) ? -ov.y * v_alpha : 0.0f;
                    zacc = fmaf(-fac, cv, zacc);
                    T = ok1 ? Tn : T;
#ifdef GS_RASTER_STATS
                    if (ok1) GS_STAT_ALL(7, 1);
                    if (has1 && !ok1) { if (idx1 > bin_final) GS_STAT2_ALL(2, 1); else GS_STAT2_ALL(3, 1); }
#endif
                    pairbuf[has1 ? e1 : GS_LOG_PAIR_CAP] = make_float2(s_out, fac);
                }
            } while (__ballot(list != 0ull) != 0ull);
            lanes_lds_sync();
#ifdef GS_RASTER_PHASES
            if (blockIdx.x == 0 && threadIdx.x == 0) { const long long _t = (long long)__builtin_readcyclecounter(); atomicAdd(&g_raster_stats[2], (unsigned long long)(_t - _pw0)); _pw0 = _t; }
#endif
            // ---- reduction: lane j sums the pairs of record j over the set bits of its pixel mask (pixel order)
            float sum[NV];
            {
                const float X = ra4.x - ((float)qx0 + 0.5f), Y = ra4.y - ((float)qy0 + 0.5f);    // dx = X - x,  dy = Y - y
                unsigned long long m = pms;
                int e = cum - cnt - cumbase;
                float m0 = 0.0f, mxy = 0.0f, c2s = 0.0f;
                v2f m1 = (v2f)(0.0f), m2 = (v2f)(0.0f), c01 = (v2f)(0.0f);
                if (__ballot(m != 0ull) != 0ull) do {                 // (rotated by hand: the while form copied all nine accumulators every trip)
                    GS_STAT2(7, 1);
#if defined(GS_RASTER_STATS) && !defined(GS_RASTER_PHASES) && defined(GS_HIST_REDUCTION)      /* bank 3 counts the REDUCTION's trips instead */
                    { const int na = __popcll(__ballot(m != 0ull));
#pragma unroll
                      for (int k = 0; k < 8; ++k) walk_hist[k] += (na > 8 * k && na <= 8 * k + 8) ? 1 : 0; }
#endif
#ifdef GS_RASTER_PHASES
                    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_raster_stats[7], 1ull);
#endif
                    const bool has = m != 0ull;
                    const int p = gs_pop_lowest(m);
                    const float2 sfr = pairbuf[has ? e : 0];
                    float4 vr;
                    vr.x = __shfl(v_rc[0], p, 64); vr.y = CD > 1 ? __shfl(v_rc[1], p, 64) : 0.0f; vr.z = CD > 2 ? __shfl(v_rc[2], p, 64) : 0.0f; vr.w = 0.0f;
                    e += has ? 1 : 0;
                    const float s_w = has ? sfr.x : 0.0f, f_w = has ? sfr.y : 0.0f;
                    const v2f d = v2f{X, Y} - v2f{(float)(p & 7), (float)(p >> 3)};
                    const v2f sd = d * s_w;
                    m0 += s_w;
                    m1 += sd;
                    m2 = __builtin_elementwise_fma(sd, d, m2);
                    mxy = fmaf(sd.x, d.y, mxy);
                    c01 = __builtin_elementwise_fma((v2f)(f_w), v2f{vr.x, vr.y}, c01);
                    if (CD > 2) c2s = fmaf(f_w, vr.z, c2s);
                } while (__ballot(m != 0ull) != 0ull);
                sum[0] = m0; sum[1] = m1.x; sum[2] = m1.y; sum[3] = m2.x; sum[4] = mxy; sum[5] = m2.y;
                sum[6] = c01.x;
                if (CD > 1) sum[7] = c01.y;
                if (CD > 2) sum[8] = c2s;
            }
            lanes_lds_sync();                                          // pairbuf is dead: its space becomes the commit staging
#ifdef GS_RASTER_PHASES
            if (blockIdx.x == 0 && threadIdx.x == 0) { const long long _t = (long long)__builtin_readcyclecounter(); atomicAdd(&g_raster_stats[3], (unsigned long long)(_t - _pw0)); _pw0 = _t; }
#endif
            {
                const float ga = ra4.z, gb = ra4.w, gc = rb4.x, go = rb4.y;
                const float M0 = sum[0], Mx = sum[1], My = sum[2];
                float out[NV];
                out[0] = (2.0f * ga) * Mx + gb * My;
                out[1] = gb * Mx + (2.0f * gc) * My;
                out[2] = 0.5f * sum[3]; out[3] = sum[4]; out[4] = 0.5f * sum[5];
                out[5] = (M0 != 0.0f) ? -M0 / go : 0.0f;             // sum of vis * v_alpha over the uncapped pairs
#pragma unroll
                for (int k = 0; k < CD; ++k) out[6 + k] = sum[6 + k];
                if (mine) {
#pragma unroll
                    for (int k = 0; k < NV; ++k) stage[lane * NV + k] = out[k];
                }
            }
            lanes_lds_sync();
            // ---- commit: RPI records x NV values per atomic instruction; the NV lanes of a record hit one packed gradient record
            {
                const int r = lane / NV, k = lane - r * NV;
                for (int it = 0; r0 + it * RPI < r1; ++it) {
                    const int j = r0 + it * RPI + r;
                    if (r < RPI && j < r1 && k < 6 + D) {
                        const float v = stage[j * NV + k];
                        if (v != 0.0f) {
                            const int g = __float_as_int(qc[j].w);
#ifndef GS_EXP_NOATOMIC
                            gs_atomic_add(v_packed + (size_t)g * rec_stride + k, v);
#else
                            if (v == 123456.0f) v_packed[(size_t)g * rec_stride + k] = v;      /* timing experiment only */
#endif
                        }
                    }
                }
            }
#ifdef GS_RASTER_PHASES
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_raster_stats[4], (unsigned long long)((long long)__builtin_readcyclecounter() - _pw0));
#endif
            cumbase = __builtin_amdgcn_readlane(cum, r1 - 1);
            r0 = r1;
        }
    }
#if defined(GS_RASTER_STATS) && !defined(GS_RASTER_PHASES)
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (walk_hist[k]) atomicAdd(&g_raster_stats3[k], (unsigned long long)walk_hist[k]);
    }
#endif
    GS_TL_END();
}

// ---------------------------------------------------------------------------------------------------
// Occupancy cap of the two compositor kernels: they use no LDS, so a dynamic LDS request of 160 KB / 4 limits a CU to
// FOUR resident 256-thread blocks (4 waves per SIMD instead of 8).  Measured on the bench workload: backward
// 1.48 -> 1.19 ms, forward 0.68 -> 0.64 ms per view (3 blocks: same; 2 blocks: back to 1.49 ms; wave priorities for
// the long tiles: no effect).  The walks are serial dependency chains bound by VALU issue; four waves per SIMD already
// saturate it, the other four only stretch every chain, delay the long tiles and thrash the scalar/vector L1s -- and
// the freed wave slots let the memory-bound kernels of the engine's other two streams co-reside.
#define GS_RASTER_BLOCKS_PER_CU 4
static int gs_raster_blocks_per_cu() { return GS_RASTER_BLOCKS_PER_CU; }
static size_t gs_raster_lds_pad() { return (size_t)(160 * 1024 / gs_raster_blocks_per_cu()) - 1024; }
// LDS of a compositor launch: what the kernel needs, padded up ONLY while one more block than the cap would still fit -- the blocks
// then leave what they do not use (forward window: 6 KB per block, 24 KB per CU) to the front-side kernels that run beside them
static size_t gs_raster_lds(size_t natural)
{
    const size_t pad = gs_raster_lds_pad();
    if (natural >= pad) return natural;
    if ((size_t)(gs_raster_blocks_per_cu() + 1) * (natural + 512) > (size_t)160 * 1024) return natural;
    return pad;
}
// workspace layout: [rec0 | rec1 | rec2 | tile_order], every segment 256-byte aligned
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t gs_raster_ws_bytes(int64_t n_isects, int V, int W, int H, int tile_size)
{
    if (tile_size <= 0) return 0;
    const size_t tiles = (size_t)((W + tile_size - 1) / tile_size) * (size_t)((H + tile_size - 1) / tile_size);
    const size_t n = n_isects > 0 ? (size_t)n_isects : 1;
    const size_t v = V > 0 ? (size_t)V : 1;
    return 3 * align256(n * sizeof(float4)) + align256(tiles * sizeof(int32_t)) + align256(64 * sizeof(int32_t)) +
           align256(64 * tiles * sizeof(int32_t)) + align256(4 * v * sizeof(float4));
}

struct RasterWs {
    float4 *rec0, *rec1, *rec2;
    int32_t* order;
    BwdOrder bo;
    float4* vis;                    // scratch of the forward only: 64-byte per-visible records
};
static RasterWs carve(void* ws, int64_t n_isects, int V, int tiles)
{
    const size_t n = n_isects > 0 ? (size_t)n_isects : 1;
    const size_t v = V > 0 ? (size_t)V : 1;
    char* p = (char*)ws;
    RasterWs r;
    r.rec0 = (float4*)p; p += align256(n * sizeof(float4));
    r.rec1 = (float4*)p; p += align256(n * sizeof(float4));
    r.rec2 = (float4*)p; p += align256(n * sizeof(float4));
    r.order = (int32_t*)p; p += align256((size_t)tiles * sizeof(int32_t));
    r.bo.bcount = (int32_t*)p; p += align256(64 * sizeof(int32_t));
    r.bo.blist = (int32_t*)p; p += align256(64 * (size_t)tiles * sizeof(int32_t));
    r.vis = (float4*)p;
    return r;
}

// Capacity protocol: while a *_cap entry point is on the stack, the kernels launched by the functions below read the actual
// intersection count from device memory (counts_dev = {V, I} as written by gs_project_fwd) and treat the scalar as a capacity.
static thread_local const long long* t_counts_dev = nullptr;
struct CountsScope {
    explicit CountsScope(const int64_t* c) { t_counts_dev = (const long long*)c; }
    ~CountsScope() { t_counts_dev = nullptr; }
};
static GsCount isect_count(int64_t n) { return GsCount{ (long long)n, t_counts_dev ? t_counts_dev + 1 : nullptr }; }
// The same mechanism for the tone-mapping variants: while a gs_raster_*_tone* entry point is on the stack the launches below hand
// its ToneFwd / ToneBwd to the kernels (zero = plain compositor).
static thread_local ToneFwd t_tone_fwd = { 0, nullptr, nullptr };
static thread_local ToneBwd t_tone_bwd = { 0, nullptr, nullptr, nullptr, nullptr };
static thread_local CullLog t_cull_log = { nullptr, nullptr, nullptr };
struct CullLogScope { explicit CullLogScope(const CullLog& l) { t_cull_log = l; } ~CullLogScope() { t_cull_log = CullLog{ nullptr, nullptr, nullptr }; } };
struct ToneFwdScope { explicit ToneFwdScope(const ToneFwd& t) { t_tone_fwd = t; } ~ToneFwdScope() { t_tone_fwd = ToneFwd{ 0, nullptr, nullptr }; } };
struct ToneBwdScope { explicit ToneBwdScope(const ToneBwd& t) { t_tone_bwd = t; } ~ToneBwdScope() { t_tone_bwd = ToneBwd{ 0, nullptr, nullptr, nullptr, nullptr }; } };

template <int CD>
static int launch_fwd(int W, int H, int D, const RasterWs& ws, const float* colors, const float* background,
                      int64_t n_isects, const int32_t* offsets, float* render, float* alphas, int32_t* last_ids,
                      hipStream_t s)
{
    const int tile_w = (W + GS_TILE - 1) / GS_TILE, tile_h = (H + GS_TILE - 1) / GS_TILE;
    if (t_tone_fwd.image && !(CD == 3 && D == 3 && background == nullptr)) {
        gs_set_error("gs_raster_composite_tone: needs D == 3 and no background");
        return GS_EINVAL;
    }
    {                                                         // sliding window of two dense batches, every D
        const size_t planes = CD > 3 ? 4 * (size_t)2 * (size_t)D * 64 * sizeof(float) : 0;      // colour planes of two batches per wave
        const size_t lds = gs_raster_lds(4 * (size_t)GS_WIN_Q_BYTES + planes);
        // (D > 14 or so: the opt-in for more than 64 KB of dynamic LDS -- set on EVERY such launch: the attribute is per device, a
        //  process may drive several GPUs and threads, so nothing about it is cached here)
        if (lds > 64 * 1024)
            GS_CHECK_HIP(hipFuncSetAttribute((const void*)raster_fwd_window_kernel<CD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(raster_fwd_window_kernel<CD>, dim3(tile_w * tile_h), dim3(256), lds, s, W, H, tile_w, tile_w * tile_h, D,
                           ws.order, ws.rec0, ws.rec1, ws.rec2, colors, background, isect_count(n_isects), offsets, render, alphas,
                           last_ids, t_tone_fwd, t_cull_log, t_cull_log.count ? ws.bo : BwdOrder{ nullptr, nullptr });
        GS_CHECK_LAUNCH();
        return GS_OK;
    }
}

static int raster_check(const char* who, int W, int H, int tile_size, int D, int V, int64_t n_isects, const void* ws,
                        size_t ws_bytes)
{
    GS_CHECK_ARG(W > 0 && H > 0, "bad image size");
    GS_CHECK_ARG(tile_size == GS_TILE, "only tile_size=16 is built (rfstudio/model/gsplat.py:30)");
    GS_CHECK_ARG(D >= 1 && D <= GS_MAX_CHANNELS, "1 <= D <= 32");
    GS_CHECK_ARG(n_isects >= 0 && n_isects < (1ll << 31), "n_isects must fit int32");
    GS_CHECK_ARG(ws != nullptr, "workspace must not be NULL");
    GS_CHECK_ARG(V >= 0, "bad V");
    if (ws_bytes < gs_raster_ws_bytes(n_isects, V, W, H, tile_size)) { gs_set_error("%s: workspace too small", who); return GS_ENOSPC; }
    return GS_OK;
}

// A5 preparation: per-visible records, the sorted record stream and the longest-first tile order (HBM-bound; a caller
// that overlaps streams runs it next to the sort, away from the VALU-bound compositor).
static int raster_prepare_impl(int W, int H, int tile_size, int D, int V, const float* means2d, const float* conics,
                               const float* opacities, const float* colors, const float* vis_records, int64_t n_isects,
                               const int32_t* offsets, const int32_t* flatten_ids, void* ws, size_t ws_bytes, void* stream);

extern "C" int gs_raster_prepare(int W, int H, int tile_size, int D, int V, const float* means2d, const float* conics,
                                 const float* opacities, const float* colors, int64_t n_isects, const int32_t* offsets,
                                 const int32_t* flatten_ids, void* ws, size_t ws_bytes, void* stream)
{
    return raster_prepare_impl(W, H, tile_size, D, V, means2d, conics, opacities, colors, nullptr, n_isects, offsets, flatten_ids,
                               ws, ws_bytes, stream);
}

extern "C" int gs_raster_prepare_vis(int W, int H, int tile_size, int D, int V, const float* vis_records, int64_t n_isects,
                                     const int32_t* offsets, const int32_t* flatten_ids, void* ws, size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(vis_records != nullptr || V == 0, "vis_records (written by gs_project_fwd_vis) must not be NULL");
    return raster_prepare_impl(W, H, tile_size, D, V, nullptr, nullptr, nullptr, nullptr, vis_records, n_isects, offsets,
                               flatten_ids, ws, ws_bytes, stream);
}

static int raster_prepare_impl(int W, int H, int tile_size, int D, int V, const float* means2d, const float* conics,
                               const float* opacities, const float* colors, const float* vis_records, int64_t n_isects,
                               const int32_t* offsets, const int32_t* flatten_ids, void* ws, size_t ws_bytes, void* stream)
{
    const int rc = raster_check("gs_raster_prepare", W, H, tile_size, D, V, n_isects, ws, ws_bytes);
    if (rc != GS_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = ((W + GS_TILE - 1) / GS_TILE) * ((H + GS_TILE - 1) / GS_TILE);
    const RasterWs r = carve(ws, n_isects, V, tiles);
    if (n_isects > 0 && V > 0) {
        const float4* vis = (const float4*)vis_records;
        if (!vis) {                                          // per-visible records not supplied by the projection: build them here
            hipLaunchKernelGGL(pack_visible_kernel, dim3(gs_cdiv(V, 256)), dim3(256), 0, s, V, D, means2d, conics, opacities, colors,
                               r.vis);
            GS_CHECK_LAUNCH();
            vis = r.vis;
        }
        hipLaunchKernelGGL(build_stream_packed_kernel, dim3(gs_cdiv(n_isects, 256)), dim3(256), 0, s, isect_count(n_isects), flatten_ids,
                           vis, r.rec0, r.rec1, r.rec2);
        GS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, tiles, isect_count(n_isects), offsets, r.order, r.bo);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// A5 proper on a prepared workspace
extern "C" int gs_raster_composite(int W, int H, int tile_size, int D, int V, const float* colors, const float* background,
                                   int64_t n_isects, const int32_t* offsets, float* render, float* alphas,
                                   int32_t* last_ids, const void* ws, size_t ws_bytes, void* stream)
{
    const int rc = raster_check("gs_raster_composite", W, H, tile_size, D, V, n_isects, ws, ws_bytes);
    if (rc != GS_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = ((W + GS_TILE - 1) / GS_TILE) * ((H + GS_TILE - 1) / GS_TILE);
    const RasterWs r = carve((void*)ws, n_isects, V, tiles);
#define GS_FWD(CD) return launch_fwd<CD>(W, H, D, r, colors, background, n_isects, offsets, render, alphas, last_ids, s)
    if (D <= 3) GS_FWD(3);
    if (D <= 4) GS_FWD(4);
    if (D <= 8) GS_FWD(8);
    if (D <= 16) GS_FWD(16);
    GS_FWD(32);
#undef GS_FWD
}

extern "C" int gs_raster_fwd(int W, int H, int tile_size, int D, int V, const float* means2d, const float* conics,
                             const float* opacities, const float* colors, const float* background,
                             int64_t n_isects, const int32_t* offsets, const int32_t* flatten_ids,
                             float* render, float* alphas, int32_t* last_ids, void* ws, size_t ws_bytes, void* stream)
{
    const int rc = gs_raster_prepare(W, H, tile_size, D, V, means2d, conics, opacities, colors, n_isects, offsets, flatten_ids,
                                     ws, ws_bytes, stream);
    if (rc != GS_OK) return rc;
    return gs_raster_composite(W, H, tile_size, D, V, colors, background, n_isects, offsets, render, alphas, last_ids, ws,
                               ws_bytes, stream);
}

template <int CD>
static int launch_bwd(int W, int H, int D, const RasterWs& ws, const float* colors, const float* background,
                      int64_t n_isects, const int32_t* offsets, const float* alphas, const int32_t* last_ids,
                      const float* v_render, const float* v_alphas, float* v_packed, int rec_stride, hipStream_t s)
{
    const int tile_w = (W + GS_TILE - 1) / GS_TILE, tile_h = (H + GS_TILE - 1) / GS_TILE;
    if (t_tone_bwd.v_image && !(CD == 3 && D == 3 && background == nullptr)) {
        gs_set_error("gs_raster_bwd_tone: needs D == 3 and no background");
        return GS_EINVAL;
    }
    if constexpr (CD <= 3) {                                  // colours travel in the record stream only for D <= 3
        if (t_cull_log.idx) {                                   // the forward left its cull log: no fill, no masks
            const size_t lds = gs_raster_lds(4 * (size_t)LogLds::WAVE_BYTES);
            hipLaunchKernelGGL(raster_bwd_log_kernel<CD>, dim3(tile_w * tile_h), dim3(256), lds, s, W, H, tile_w, tile_w * tile_h, D,
                               ws.order, ws.rec0, ws.rec1, ws.rec2, background, isect_count(n_isects), offsets, alphas, last_ids,
                               v_render, v_alphas, v_packed, rec_stride, t_tone_bwd, t_cull_log, ws.bo);
            GS_CHECK_LAUNCH();
            return GS_OK;
        }
    }
    {                                                         // no log (gs_raster_bwd / _acc / _tone_acc), and every D > 3: pair buffer, own cull + masks
        size_t lds = 4 * ((size_t)Lanes2Lds<CD>::WAVE_BYTES + (CD > 3 ? 2 * (size_t)D * 64 * 4 : 0));      // (+ colour and cotangent planes)
        if (lds < gs_raster_lds_pad()) lds = gs_raster_lds_pad();
        if (lds > 64 * 1024)                           // > 64 KB of dynamic LDS needs the opt-in (per device: set on every such launch)
            GS_CHECK_HIP(hipFuncSetAttribute((const void*)raster_bwd_lanes2_kernel<CD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(raster_bwd_lanes2_kernel<CD>, dim3(tile_w * tile_h), dim3(256), lds, s, W, H, tile_w, tile_w * tile_h, D,
                           ws.order, ws.rec0, ws.rec1, ws.rec2, colors, background, isect_count(n_isects), offsets, alphas, last_ids,
                           v_render, v_alphas, v_packed, rec_stride, t_tone_bwd);
        GS_CHECK_LAUNCH();
        return GS_OK;
    }
}

extern "C" int gs_raster_grad_stride(int D) { return ((6 + D) + 15) / 16 * 16; }

static int raster_bwd_impl(int W, int H, int tile_size, int D, int V, const float* colors, const float* background,
                           int64_t n_isects, const int32_t* offsets, const float* alphas, const int32_t* last_ids,
                           const float* v_render, const float* v_alphas, float* v_packed, const void* ws,
                           size_t ws_bytes, void* stream, bool zero);

extern "C" int gs_raster_bwd(int W, int H, int tile_size, int D, int V, const float* colors, const float* background,
                             int64_t n_isects, const int32_t* offsets, const float* alphas, const int32_t* last_ids,
                             const float* v_render, const float* v_alphas, float* v_packed, const void* ws,
                             size_t ws_bytes, void* stream)
{
    return raster_bwd_impl(W, H, tile_size, D, V, colors, background, n_isects, offsets, alphas, last_ids, v_render, v_alphas,
                           v_packed, ws, ws_bytes, stream, true);
}

extern "C" int gs_raster_bwd_acc(int W, int H, int tile_size, int D, int V, const float* colors, const float* background,
                                 int64_t n_isects, const int32_t* offsets, const float* alphas, const int32_t* last_ids,
                                 const float* v_render, const float* v_alphas, float* v_packed, const void* ws,
                                 size_t ws_bytes, void* stream)
{
    return raster_bwd_impl(W, H, tile_size, D, V, colors, background, n_isects, offsets, alphas, last_ids, v_render, v_alphas,
                           v_packed, ws, ws_bytes, stream, false);
}

static int raster_bwd_impl(int W, int H, int tile_size, int D, int V, const float* colors, const float* background,
                           int64_t n_isects, const int32_t* offsets, const float* alphas, const int32_t* last_ids,
                           const float* v_render, const float* v_alphas, float* v_packed, const void* ws,
                           size_t ws_bytes, void* stream, bool zero)
{
    GS_CHECK_ARG(W > 0 && H > 0 && V >= 0, "bad sizes");
    GS_CHECK_ARG(tile_size == GS_TILE, "only tile_size=16 is built (rfstudio/model/gsplat.py:30)");
    GS_CHECK_ARG(D >= 1 && D <= GS_MAX_CHANNELS, "1 <= D <= 32");
    GS_CHECK_ARG(n_isects >= 0 && n_isects < (1ll << 31), "n_isects must fit int32");
    GS_CHECK_ARG(ws != nullptr, "workspace (the stream written by gs_raster_fwd) must not be NULL");
    if (ws_bytes < gs_raster_ws_bytes(n_isects, V, W, H, tile_size)) { gs_set_error("gs_raster_bwd: workspace too small"); return GS_ENOSPC; }
    hipStream_t s = (hipStream_t)stream;
    const int rec_stride = gs_raster_grad_stride(D);
    if (V > 0 && zero) GS_CHECK_HIP(gs_zero_async(v_packed, sizeof(float) * (size_t)rec_stride * (size_t)V, s));
    if (n_isects == 0 || V == 0) return GS_OK;
    const int tiles = ((W + GS_TILE - 1) / GS_TILE) * ((H + GS_TILE - 1) / GS_TILE);
    const RasterWs r = carve((void*)ws, n_isects, V, tiles);
#define GS_BWD(CD) return launch_bwd<CD>(W, H, D, r, colors, background, n_isects, offsets, alphas, last_ids, v_render, v_alphas, v_packed, rec_stride, s)
    if (D <= 3) GS_BWD(3);
    if (D <= 4) GS_BWD(4);
    if (D <= 8) GS_BWD(8);
    if (D <= 16) GS_BWD(16);
    GS_BWD(32);
#undef GS_BWD
}

// ---------------------------------------------------------------------------------------------------
// Capacity-protocol variants (include/geosplat_hip.h): V_cap / n_isects_cap size the workspace and the grids, the actual counts
// are read on the device from counts_dev = {V, I}; nothing here needs the host to know them.
extern "C" int gs_raster_prepare_vis_cap(int W, int H, int tile_size, int D, int V_cap, const float* vis_records, int64_t n_isects_cap,
                                         const int64_t* counts_dev, const int32_t* offsets, const int32_t* flatten_ids, void* ws,
                                         size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(counts_dev != nullptr, "counts_dev must not be NULL");
    CountsScope sc(counts_dev);
    return gs_raster_prepare_vis(W, H, tile_size, D, V_cap, vis_records, n_isects_cap, offsets, flatten_ids, ws, ws_bytes, stream);
}

extern "C" int gs_raster_composite_cap(int W, int H, int tile_size, int D, int V_cap, const float* colors, const float* background,
                                       int64_t n_isects_cap, const int64_t* counts_dev, const int32_t* offsets, float* render,
                                       float* alphas, int32_t* last_ids, const void* ws, size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(counts_dev != nullptr, "counts_dev must not be NULL");
    CountsScope sc(counts_dev);
    return gs_raster_composite(W, H, tile_size, D, V_cap, colors, background, n_isects_cap, offsets, render, alphas, last_ids, ws,
                               ws_bytes, stream);
}

extern "C" int gs_raster_bwd_cap(int W, int H, int tile_size, int D, int V_cap, const float* colors, const float* background,
                                 int64_t n_isects_cap, const int64_t* counts_dev, const int32_t* offsets, const float* alphas,
                                 const int32_t* last_ids, const float* v_render, const float* v_alphas, float* v_packed, const void* ws,
                                 size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(counts_dev != nullptr, "counts_dev must not be NULL");
    CountsScope sc(counts_dev);
    return gs_raster_bwd(W, H, tile_size, D, V_cap, colors, background, n_isects_cap, offsets, alphas, last_ids, v_render, v_alphas,
                         v_packed, ws, ws_bytes, stream);
}

extern "C" int gs_raster_bwd_acc_cap(int W, int H, int tile_size, int D, int V_cap, const float* colors, const float* background,
                                     int64_t n_isects_cap, const int64_t* counts_dev, const int32_t* offsets, const float* alphas,
                                     const int32_t* last_ids, const float* v_render, const float* v_alphas, float* v_packed,
                                     const void* ws, size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(counts_dev != nullptr, "counts_dev must not be NULL");
    CountsScope sc(counts_dev);
    return gs_raster_bwd_acc(W, H, tile_size, D, V_cap, colors, background, n_isects_cap, offsets, alphas, last_ids, v_render, v_alphas,
                             v_packed, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------
// Compositor with S4 inside (see ToneFwd / ToneBwd).  The plain outputs (render, alphas, last_ids) are written as well: the backward
// needs them.  n_isects / V are exact counts, or capacities when counts_dev != NULL (capacity protocol, as the *_cap entry points).
extern "C" int gs_raster_composite_tone(int W, int H, int tile_size, int V, const float* colors, int64_t n_isects,
                                        const int64_t* counts_dev, const int32_t* offsets, float* render, float* alphas,
                                        int32_t* last_ids, int tone_mode, const float* exposure, float* image, const void* ws,
                                        size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(tone_mode >= 0 && tone_mode <= 2 && exposure != nullptr && image != nullptr, "bad tone mode / exposure / image");
    CountsScope sc(counts_dev);
    ToneFwdScope ts(ToneFwd{ tone_mode, exposure, (float4*)image });
    return gs_raster_composite(W, H, tile_size, 3, V, colors, nullptr, n_isects, offsets, render, alphas, last_ids, ws, ws_bytes, stream);
}

// v_packed is ACCUMULATED into (as gs_raster_bwd_acc), v_exposure too (one atomic per quadrant wave).
extern "C" int gs_raster_bwd_tone_acc(int W, int H, int tile_size, int V, const float* colors, int64_t n_isects,
                                      const int64_t* counts_dev, const int32_t* offsets, const float* render, const float* alphas,
                                      const int32_t* last_ids, int tone_mode, const float* exposure, const float* v_image,
                                      float* v_packed, float* v_exposure, const void* ws, size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(tone_mode >= 0 && tone_mode <= 2 && exposure != nullptr && v_image != nullptr && render != nullptr && v_exposure != nullptr,
                 "bad tone mode / exposure / v_image / render / v_exposure");
    CountsScope sc(counts_dev);
    ToneBwdScope ts(ToneBwd{ tone_mode, exposure, render, (const float4*)v_image, v_exposure });
    return gs_raster_bwd_acc(W, H, tile_size, 3, V, colors, nullptr, n_isects, offsets, alphas, last_ids, nullptr, nullptr, v_packed, ws,
                             ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------
// self-test hook: counts the floats with bit patterns in [lo_bits, hi_bits] for which gs_rcp_exact2 differs from IEEE division
// ---- cull log (forward -> backward), see struct CullLog -----------------------------------------------------------------------
static CullLog carve_log(void* log_ws, int64_t n_isects)
{
    const size_t n = n_isects > 0 ? (size_t)n_isects : 1;
    char* p = (char*)log_ws;
    CullLog l;
    l.mask = (unsigned long long*)p; p += align256(4 * n * sizeof(unsigned long long));
    l.idx = (int32_t*)p; p += align256(4 * n * sizeof(int32_t));
    l.count = (int32_t*)p;
    return l;
}
extern "C" size_t gs_raster_log_ws_bytes(int64_t n_isects, int W, int H, int tile_size)
{
    if (tile_size <= 0) return 0;
    const size_t n = n_isects > 0 ? (size_t)n_isects : 1;
    const size_t tiles = (size_t)((W + tile_size - 1) / tile_size) * (size_t)((H + tile_size - 1) / tile_size);
    return align256(4 * n * sizeof(unsigned long long)) + align256(4 * n * sizeof(int32_t)) + align256(4 * tiles * sizeof(int32_t));
}

extern "C" int gs_raster_composite_tone_log(int W, int H, int tile_size, int V, const float* colors, int64_t n_isects,
                                            const int64_t* counts_dev, const int32_t* offsets, float* render, float* alphas,
                                            int32_t* last_ids, int tone_mode, const float* exposure, float* image, const void* ws,
                                            size_t ws_bytes, void* log_ws, size_t log_bytes, void* stream)
{
    GS_CHECK_ARG(log_ws != nullptr && tile_size == GS_TILE, "log_ws must not be NULL");
    if (log_bytes < gs_raster_log_ws_bytes(n_isects, W, H, tile_size)) { gs_set_error("gs_raster_composite_tone_log: log workspace too small"); return GS_ENOSPC; }
    CullLogScope ls(carve_log(log_ws, n_isects));
    return gs_raster_composite_tone(W, H, tile_size, V, colors, n_isects, counts_dev, offsets, render, alphas, last_ids, tone_mode, exposure,
                                    image, ws, ws_bytes, stream);
}

extern "C" int gs_raster_bwd_tone_log_acc(int W, int H, int tile_size, int V, const float* colors, int64_t n_isects,
                                          const int64_t* counts_dev, const int32_t* offsets, const float* render, const float* alphas,
                                          const int32_t* last_ids, int tone_mode, const float* exposure, const float* v_image,
                                          float* v_packed, float* v_exposure, const void* ws, size_t ws_bytes, const void* log_ws,
                                          size_t log_bytes, void* stream)
{
    GS_CHECK_ARG(log_ws != nullptr && tile_size == GS_TILE, "log_ws must not be NULL");
    if (log_bytes < gs_raster_log_ws_bytes(n_isects, W, H, tile_size)) { gs_set_error("gs_raster_bwd_tone_log_acc: log workspace too small"); return GS_ENOSPC; }
    CullLogScope ls(carve_log((void*)log_ws, n_isects));
    return gs_raster_bwd_tone_acc(W, H, tile_size, V, colors, n_isects, counts_dev, offsets, render, alphas, last_ids, tone_mode, exposure,
                                  v_image, v_packed, v_exposure, ws, ws_bytes, stream);
}

__global__ void __launch_bounds__(256) selftest_rcp_kernel(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* mismatches)
{
    const uint64_t n = (uint64_t)hi_bits - lo_bits + 1;
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        const v2f r = gs_rcp_exact2(v2f{x, x});
        const float d = __fdiv_rn(1.0f, x);
        bad += (__float_as_uint(r.x) != __float_as_uint(d)) + (__float_as_uint(r.y) != __float_as_uint(d));
    }
    if (bad) atomicAdd(mismatches, bad);
}

extern "C" int gs_selftest_rcp(uint32_t lo_bits, uint32_t hi_bits, uint64_t* mismatches_dev, void* stream)
{
    GS_CHECK_ARG(mismatches_dev != nullptr && hi_bits >= lo_bits, "bad range");
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_HIP(gs_zero_async(mismatches_dev, sizeof(uint64_t), s));
    hipLaunchKernelGGL(selftest_rcp_kernel, dim3(4096), dim3(256), 0, s, lo_bits, hi_bits, (unsigned long long*)mismatches_dev);
    GS_CHECK_LAUNCH();
    return GS_OK;
}

// self-test hook for the canonical exponential (gs_common.h gs_exp_neg): over every float with bit pattern in [lo_bits, hi_bits]
//   out[0] = sum_i bits(gs_exp_neg(sigma_i)) * (2 i + 1) mod 2^64   (the oracle's gso_exp_neg_check computes the same sum:
//            equal sums <=> the two implementations agree bit for bit on the whole range)
//   out[1] = bits of the largest relative error against the float64 exponential (a non-negative double orders like its bits)
__global__ void __launch_bounds__(256) selftest_exp_kernel(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* out)
{
    const uint64_t n = (uint64_t)hi_bits - lo_bits + 1;
    unsigned long long sum = 0;
    double worst = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        const float r = gs_exp_neg(x);
        sum += (unsigned long long)__float_as_uint(r) * (2ull * i + 1ull);
        const double ref = exp(-(double)x);
        const double e = fabs((double)r - ref) / ref;
        worst = e > worst ? e : worst;
    }
    atomicAdd(out, sum);
    atomicMax(out + 1, (unsigned long long)__double_as_longlong(worst));
}

extern "C" int gs_selftest_exp(uint32_t lo_bits, uint32_t hi_bits, uint64_t* out_dev, void* stream)
{
    GS_CHECK_ARG(out_dev != nullptr && hi_bits >= lo_bits, "bad range");
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_HIP(gs_zero_async(out_dev, 2 * sizeof(uint64_t), s));
    hipLaunchKernelGGL(selftest_exp_kernel, dim3(4096), dim3(256), 0, s, lo_bits, hi_bits, (unsigned long long*)out_dev);
    GS_CHECK_LAUNCH();
    return GS_OK;
}
