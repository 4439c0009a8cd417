// gs_sort.hip -- A2 + A3: tile binning and the stable (tile | depth) order of the intersections, hand-written for gfx950.
// Semantics: gsplat 1.4 `isect_tiles(sort=True)` as reached from rfstudio/model/gsplat.py:334 -- keys
// (tile_id << 32) | float_bits(depth), payload = packed Gaussian index, stable ascending radix sort, so that equal keys
// keep their emission order (= ascending packed index); that order fixes the composited order bit-exactly.
//
// Two entry points, both on the radix machinery below (no vendor sort library on the path):
//   gs_isect_sort : the upstream call shape -- sorts already emitted (isect_ids, flatten_ids) pairs, six 8-bit passes
//                   over the 32 + tile_bits significant key bits (kept for callers that hold emitted keys);
//   gs_isect_bin  : what rasterization() and the engine use.  The 44-bit key of an intersection is (tile, depth of
//                   its Gaussian): the depth half is a property of the V Gaussians, not of the I = 2.1 V intersections,
//                   so it is sorted ONCE on the Gaussians (4 passes over 8-byte (depth, index) items, V-sized), the
//                   intersections are then EMITTED in depth order as 8-byte (tile, index) items, and two stable passes
//                   over the tile bits (6 + 6 for 2 500 tiles) finish the order: 2 I-sized passes of 8-byte items instead
//                   of 6 passes of 12-byte pairs, and the unsorted key array is never materialised.  Stability of every
//                   pass makes the result identical to the upstream order: by tile, then depth bits, then packed index.
//
// One radix pass = three kernels: per-block digit histogram -> exclusive scan of the [digit][block] table -> scatter.
// A block owns 4 096 consecutive items, a wave 1 024 of them (striped over its lanes: coalesced, and wave-major order =
// index order), and ranks them with ballots: for every item the lanes holding the same digit are found with one ballot
// per digit bit, the rank inside the wave is a popcount, and one lane per digit group advances the wave's running
// counter in LDS -- no sorting network, no atomics.
#include "gs_common.h"

#include <cstdlib>
#include <cstring>

#define RS_THREADS 256
#define RS_WAVES 4
#ifndef RS_ITEMS
#define RS_ITEMS 8                        // per thread.  16 until the end of round 4: alone the passes take the same time, inside the
#endif                                    // step the 22 KB blocks find room beside the compositor's (39 KB blocks did not): 659 -> 669
                                          // views/s (4: 664, 12: 662)
#define RS_TILE (RS_THREADS * RS_ITEMS)   // 2048 items per block
#define RS_WCHUNK (64 * RS_ITEMS)         // 512 items per wave

typedef unsigned long long u64;

static int tile_bits(int tile_w, int tile_h)
{
    unsigned n = (unsigned)(tile_w * tile_h);       // floor(log2(n_tiles)) + 1, as upstream
    int b = 0;
    while (n > 1) { n >>= 1; ++b; }
    return b + 1;
}

// ---- item kinds ----------------------------------------------------------------------------------------------------
struct PairItem { u64 key; int32_t val; };                      // emitted (isect_id, flatten_id)
struct PairIn {
    const u64* keys; const int32_t* vals;
    __device__ __forceinline__ PairItem load(int64_t i) const { return PairItem{ keys[i], vals[i] }; }
};
struct PairOut {
    u64* keys; int32_t* vals;
    __device__ __forceinline__ void store(int64_t i, const PairItem& it) const { keys[i] = it.key; vals[i] = it.val; }
};
struct PairDigit {
    int shift; unsigned mask;
    __device__ __forceinline__ unsigned operator()(const PairItem& it) const { return (unsigned)(it.key >> shift) & mask; }
};

struct U2In {
    const uint2* p;
    __device__ __forceinline__ uint2 load(int64_t i) const { return p[i]; }
};
struct U2Out {
    uint2* p;
    __device__ __forceinline__ void store(int64_t i, const uint2& it) const { p[i] = it; }
};
struct DepthIn {                                                   // (depth bits, packed index) straight from the depth array
    const float* depths;
    __device__ __forceinline__ uint2 load(int64_t i) const { return make_uint2(__float_as_uint(depths[i]), (unsigned)i); }
};
struct XDigit {                                                    // digit of .x
    int shift; unsigned mask;
    __device__ __forceinline__ unsigned operator()(const uint2& it) const { return (it.x >> shift) & mask; }
};
struct FinalTiles {                                                // last tile pass without the 64-bit keys: (tile id, flatten id) as two int32 arrays
    int32_t* tile_ids; int32_t* flatten_ids;
    __device__ __forceinline__ void store(int64_t i, const uint2& it) const
    {
        flatten_ids[i] = (int32_t)it.y;
        tile_ids[i] = (int32_t)it.x;
    }
};
struct FinalOut {                                                  // last tile pass: the two sorted meta arrays
    u64* isect_ids; int32_t* flatten_ids; const float* depths;
    __device__ __forceinline__ void store(int64_t i, const uint2& it) const
    {
        flatten_ids[i] = (int32_t)it.y;
        isect_ids[i] = ((u64)it.x << 32) | (u64)__float_as_uint(depths[it.y]);
    }
};

// ---- pass kernels --------------------------------------------------------------------------------------------------
template <typename Item, typename In, typename Digit>
__global__ void __launch_bounds__(RS_THREADS)
radix_hist_kernel(GsCount nc, In in, Digit digit, int nbins, int nblocks, unsigned* __restrict__ table)
{
    const int64_t n = gs_count(nc);
    __shared__ unsigned hist[256];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int64_t i = base + (int64_t)k * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&hist[digit(in.load(i))], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < nbins) table[(size_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// Offsets of a pass: table[d][b] (items of digit d in block b) -> exclusive prefix along b inside every digit row (one workgroup
// per digit, coalesced row) and the row totals; the scatter blocks scan the (<= 256) row totals themselves and add the two.  (A single workgroup scanning the whole 260 k-entry table was a 50 us latency chain per pass.)
__global__ void __launch_bounds__(256)
radix_rowscan_kernel(int nblocks, unsigned* __restrict__ table, unsigned* __restrict__ row_total)
{
    __shared__ unsigned wsum[4];
    unsigned* row = table + (size_t)blockIdx.x * nblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned carry = 0u;
    for (int i0 = 0; i0 < nblocks; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const unsigned v = i < nblocks ? row[i] : 0u;
        unsigned incl = v;
        incl = gs_wave_incl_scan(incl);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned before = 0u, all = 0u;
        for (int w = 0; w < 4; ++w) { if (w < wave) before += wsum[w]; all += wsum[w]; }
        if (i < nblocks) row[i] = carry + before + incl - v;
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}

// Scatter of one pass.  Ranks come from ballots (see the file header); the items are then staged through LDS in their
// block-local sorted order, so that the global stores of a digit run are contiguous (a wave writing 64 items of 64 different
// digits straight from registers is 64 partial-sector stores).
template <typename Item, typename In, typename Digit, typename Out, int NBITS>
__global__ void __launch_bounds__(RS_THREADS)
radix_scatter_kernel(GsCount nc, In in, Digit digit, int nblocks, const unsigned* __restrict__ table,
                     const unsigned* __restrict__ row_total, Out out)
{
    const int64_t n = gs_count(nc);
    constexpr int NB = 1 << NBITS;
    __shared__ unsigned cnt[RS_WAVES][NB];
    __shared__ unsigned dstart[NB];                              // start of digit d inside the block's sorted order
    __shared__ unsigned gbase[NB];                               // global position of that start
    __shared__ Item stage[RS_TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_WAVES * NB; i += RS_THREADS) (&cnt[0][0])[i] = 0u;
    __syncthreads();
    const int64_t bbase = (int64_t)blockIdx.x * RS_TILE;
    const int64_t wbase = bbase + (int64_t)wave * RS_WCHUNK;
    const int n_here = (n - bbase) < RS_TILE ? (int)((n - bbase) > 0 ? (n - bbase) : 0) : RS_TILE;
    const u64 lane_lt = (1ull << lane) - 1ull;
    Item item[RS_ITEMS];
    unsigned dig[RS_ITEMS];
    unsigned rank[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int64_t i = wbase + (int64_t)k * 64 + lane;
        const bool valid = i < n;
        if (valid) item[k] = in.load(i);
        const unsigned d = valid ? digit(item[k]) : 0u;
        u64 m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < NBITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const unsigned old = cnt[wave][d];                       // running count of this wave for digit d (same for the group)
        __builtin_amdgcn_wave_barrier();
        const unsigned r = __popcll(m & lane_lt);
        if (valid && r == 0u) cnt[wave][d] = old + (unsigned)__popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        dig[k] = d;
        rank[k] = old + r;
    }
    __syncthreads();
    // block-local start of every digit (exclusive scan of the digit totals by the first NB threads) and its global position
    {
        const int d = threadIdx.x;
        unsigned tot = 0u;
        if (d < NB) for (int w = 0; w < RS_WAVES; ++w) tot += cnt[w][d];
        unsigned incl = tot;
        incl = gs_wave_incl_scan(incl);
        __shared__ unsigned wtot[RS_WAVES];
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        unsigned before = 0u;
        for (int w = 0; w < wave; ++w) before += wtot[w];
        if (d < NB) dstart[d] = before + incl - tot;
        // global start of digit d = exclusive prefix of the row totals (256 values: every block scans them itself, which
        // saves a kernel per pass) + this block's offset inside the row
        __syncthreads();
        const unsigned rt = d < NB ? row_total[d] : 0u;
        unsigned rincl = rt;
        rincl = gs_wave_incl_scan(rincl);
        if (lane == 63) wtot[wave] = rincl;
        __syncthreads();
        unsigned rbefore = 0u;
        for (int w = 0; w < wave; ++w) rbefore += wtot[w];
        if (d < NB) gbase[d] = rbefore + rincl - rt + table[(size_t)d * nblocks + blockIdx.x];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int64_t i = wbase + (int64_t)k * 64 + lane;
        if (i < n) {
            const unsigned d = dig[k];
            unsigned off = dstart[d] + rank[k];
            for (int w = 0; w < wave; ++w) off += cnt[w][d];
            stage[off] = item[k];
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < RS_ITEMS; ++k) {
        const int p = k * RS_THREADS + (int)threadIdx.x;
        if (p < n_here) {
            const Item it = stage[p];
            const unsigned d = digit(it);
            out.store((int64_t)gbase[d] + (int64_t)(p - (int)dstart[d]), it);
        }
    }
}

template <typename Item, typename In, typename Digit, typename Out>
static int radix_pass(GsCount nc, In in, Digit digit, Out out, int nbits, unsigned* table, hipStream_t s)
{
    const int64_t n = nc.n;                                      // capacity: the grid and the table rows are sized by it
    const int nblocks = (int)((n + RS_TILE - 1) / RS_TILE);
    const int nbins = 1 << nbits;
    unsigned* row_total = table + (size_t)256 * nblocks;        // [256] + [256] behind the table
    hipLaunchKernelGGL((radix_hist_kernel<Item, In, Digit>), dim3(nblocks), dim3(RS_THREADS), 0, s, nc, in, digit, nbins, nblocks, table);
    GS_CHECK_LAUNCH();
    hipLaunchKernelGGL(radix_rowscan_kernel, dim3(nbins), dim3(256), 0, s, nblocks, table, row_total);
    GS_CHECK_LAUNCH();
    switch (nbits) {
#define RS_CASE(B) case B: hipLaunchKernelGGL((radix_scatter_kernel<Item, In, Digit, Out, B>), dim3(nblocks), dim3(RS_THREADS), 0, s, nc, in, digit, nblocks, table, row_total, out); break;
        RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7) RS_CASE(8)
#undef RS_CASE
        default: gs_set_error("radix_pass: bad digit width %d", nbits); return GS_EINVAL;
    }
    GS_CHECK_LAUNCH();
    return GS_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t table_bytes(int64_t n) { return align256(((size_t)256 * (size_t)((n + RS_TILE - 1) / RS_TILE + 1) + 512) * sizeof(unsigned)); }

// ---- gs_isect_sort: the upstream call shape ---------------------------------------------------------------------
extern "C" size_t gs_sort_ws_bytes(int64_t n_isects, int tile_w, int tile_h)
{
    if (n_isects <= 0) return 0;
    (void)tile_w; (void)tile_h;
    return table_bytes(n_isects) + align256((size_t)n_isects * 8) + align256((size_t)n_isects * 4) + 256;
}

extern "C" int gs_isect_sort(int64_t n_isects, const int64_t* isect_ids, const int32_t* flatten_ids,
                             int64_t* isect_ids_sorted, int32_t* flatten_ids_sorted, int tile_w, int tile_h,
                             void* ws, size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(n_isects >= 0 && tile_w > 0 && tile_h > 0, "bad sizes");
    if (n_isects == 0) return GS_OK;
    GS_CHECK_ARG(ws != nullptr, "workspace must not be NULL");
    if (ws_bytes < gs_sort_ws_bytes(n_isects, tile_w, tile_h)) { gs_set_error("gs_isect_sort: workspace too small"); return GS_ENOSPC; }
    hipStream_t s = (hipStream_t)stream;
    char* p = (char*)ws;
    unsigned* table = (unsigned*)p; p += table_bytes(n_isects);
    u64* tk = (u64*)p; p += align256((size_t)n_isects * 8);
    int32_t* tv = (int32_t*)p;
    const int end_bit = 32 + tile_bits(tile_w, tile_h);
    const int npass = (end_bit + 7) / 8;
    // ping-pong so that the LAST pass lands in the caller's output: pass parity decides where the first one goes
    const u64* src_k = (const u64*)isect_ids; const int32_t* src_v = flatten_ids;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = pass * 8;
        const int nbits = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
        const bool to_out = ((npass - 1 - pass) % 2) == 0;
        u64* dk = to_out ? (u64*)isect_ids_sorted : tk;
        int32_t* dv = to_out ? flatten_ids_sorted : tv;
        const int rc = radix_pass<PairItem>(GsCount{ n_isects, nullptr }, PairIn{ src_k, src_v }, PairDigit{ shift, (1u << nbits) - 1u }, PairOut{ dk, dv }, nbits, table, s);
        if (rc != GS_OK) return rc;
        src_k = dk; src_v = dv;
    }
    return GS_OK;
}

// ---- gs_isect_bin: depth-major binning --------------------------------------------------------------------------
// emission in depth order.  tile_range_exact == gs_project.hip (same operation order: the counts come from there)
__device__ __forceinline__ void tile_range_sorted(float mx, float my, int radius, int tile_size, int tw, int th,
                                                  int& x0, int& y0, int& x1, int& y1)
{
#pragma clang fp contract(off)
    const float ts = (float)tile_size;
    const float tr = (float)radius / ts;
    const float tx = mx / ts, ty = my / ts;
    const float fx0 = floorf(tx - tr), fy0 = floorf(ty - tr);
    const float fx1 = ceilf(tx + tr), fy1 = ceilf(ty + tr);
    x0 = fx0 < 0.0f ? 0 : (fx0 > (float)tw ? tw : (int)fx0);
    y0 = fy0 < 0.0f ? 0 : (fy0 > (float)th ? th : (int)fy0);
    x1 = fx1 < 0.0f ? 0 : (fx1 > (float)tw ? tw : (int)fx1);
    y1 = fy1 < 0.0f ? 0 : (fy1 > (float)th ? th : (int)fy1);
}

#define EM_THREADS 256
#ifndef EM_PER
#define EM_PER 4
#define EM_LOOK 4                            // x 64 predecessors per look-back round
#endif
#define EM_TILE (EM_THREADS * EM_PER)        // ranks per block

// tile rectangle of every visible Gaussian, packed (x0 | y0 << 16, x1 | y1 << 16): computed once in packed order (coalesced) so
// that the two depth-order kernels below gather ONE 8-byte word per Gaussian instead of means2d + radii + tiles_per_gauss
__global__ void __launch_bounds__(256)
tile_rect_kernel(GsCount vc, const float* __restrict__ means2d, const int32_t* __restrict__ radii, int tile_size, int tile_w, int tile_h,
                 uint2* __restrict__ rect)
{
    const int V = (int)gs_count(vc);
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float2 m = *reinterpret_cast<const float2*>(means2d + 2 * (size_t)v);
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    const int r = radii[v];
    if (r > 0) tile_range_sorted(m.x, m.y, r, tile_size, tile_w, tile_h, x0, y0, x1, y1);
    rect[v] = make_uint2((unsigned)x0 | ((unsigned)y0 << 16), (unsigned)x1 | ((unsigned)y1 << 16));
}
__device__ __forceinline__ unsigned rect_count(uint2 q)
{
    const int w = (int)(q.y & 0xffffu) - (int)(q.x & 0xffffu), h = (int)(q.y >> 16) - (int)(q.x >> 16);
    return (w > 0 && h > 0) ? (unsigned)(w * h) : 0u;
}

static size_t bf_state_bytes(int V) { return align256(16 + 2 * (size_t)((V + EM_TILE - 1) / EM_TILE + 1) * sizeof(u64)); }

extern "C" size_t gs_isect_bin_ws_bytes(int V, int64_t n_isects, int tile_w, int tile_h)
{
    (void)tile_w; (void)tile_h;
    const size_t v = V > 0 ? (size_t)V : 1, n = n_isects > 0 ? (size_t)n_isects : 1;
    const size_t tb = table_bytes((int64_t)(v > n ? v : n));
    return tb + 3 * align256(v * 8) + bf_state_bytes((int)v) + 2 * align256(n * 8) + 256;
}

// status word of the capacity protocol: {code, required n_isects}; written only on overflow (the caller zeroes it once)
__global__ void capacity_check_kernel(const long long* __restrict__ counts, long long v_cap, long long i_cap, long long* __restrict__ status)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long V = counts[0], I = counts[1];
    if (V > v_cap || I > i_cap) {
        status[0] = GS_ENOSPC;
        if (I > status[1]) status[1] = I;
        if (V > status[2]) status[2] = V;
    }
}

static int isect_bin_impl(int V, const float* means2d, const int32_t* radii, const float* depths, int64_t n_isects,
                          const long long* counts_dev, int tile_size, int tile_w, int tile_h, int64_t* isect_ids_sorted,
                          int32_t* flatten_ids_sorted, void* ws, size_t ws_bytes, void* stream, int32_t* tile_ids_sorted);

extern "C" int gs_isect_bin(int V, const float* means2d, const int32_t* radii, const float* depths,
                            const int32_t* tiles_per_gauss, int64_t n_isects, int tile_size, int tile_w, int tile_h,
                            int64_t* isect_ids_sorted, int32_t* flatten_ids_sorted, void* ws, size_t ws_bytes, void* stream)
{
    (void)tiles_per_gauss;                                  // the counts are re-derived from the rectangles (same arithmetic as gs_project_fwd)
    return isect_bin_impl(V, means2d, radii, depths, n_isects, nullptr, tile_size, tile_w, tile_h, isect_ids_sorted, flatten_ids_sorted,
                          ws, ws_bytes, stream, nullptr);
}

extern "C" int gs_isect_bin_cap(int V_cap, const float* means2d, const int32_t* radii, const float* depths, const int64_t* counts_dev,
                                int64_t n_isects_cap, int tile_size, int tile_w, int tile_h, int64_t* isect_ids_sorted,
                                int32_t* flatten_ids_sorted, void* ws, size_t ws_bytes, int64_t* status_dev, void* stream)
{
    GS_CHECK_ARG(counts_dev != nullptr && status_dev != nullptr, "counts_dev / status_dev must not be NULL");
    hipLaunchKernelGGL(capacity_check_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const long long*)counts_dev, (long long)V_cap,
                       (long long)n_isects_cap, (long long*)status_dev);
    GS_CHECK_LAUNCH();
    return isect_bin_impl(V_cap, means2d, radii, depths, n_isects_cap, (const long long*)counts_dev, tile_size, tile_w, tile_h,
                          isect_ids_sorted, flatten_ids_sorted, ws, ws_bytes, stream, nullptr);
}

// The same without the 64-bit `isect_ids`: a caller that only composites (the step engine: `meta` is never returned) needs the sorted
// flatten ids and the tile offsets; the last pass then writes 8 bytes per intersection instead of 12 and does not gather the depths
// (gs_isect_offsets_tiles_cap takes the int32 tile ids).
extern "C" int gs_isect_bin_tiles_cap(int V_cap, const float* means2d, const int32_t* radii, const float* depths, const int64_t* counts_dev,
                                      int64_t n_isects_cap, int tile_size, int tile_w, int tile_h, int32_t* tile_ids_sorted,
                                      int32_t* flatten_ids_sorted, void* ws, size_t ws_bytes, int64_t* status_dev, void* stream)
{
    GS_CHECK_ARG(counts_dev != nullptr && status_dev != nullptr && tile_ids_sorted != nullptr, "counts_dev / status_dev / tile_ids_sorted must not be NULL");
    hipLaunchKernelGGL(capacity_check_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const long long*)counts_dev, (long long)V_cap,
                       (long long)n_isects_cap, (long long*)status_dev);
    GS_CHECK_LAUNCH();
    return isect_bin_impl(V_cap, means2d, radii, depths, n_isects_cap, (const long long*)counts_dev, tile_size, tile_w, tile_h,
                          nullptr, flatten_ids_sorted, ws, ws_bytes, stream, tile_ids_sorted);
}

// ---- gs_isect_bin_front: binning of the engine's fused front (round 4) ---------------------------------------------------------
// Same result as gs_isect_bin_tiles_cap + gs_isect_offsets_tiles_cap -- `flatten_ids` in (tile, depth bits, packed index) order and the
// per-tile offsets, bit for bit -- from what gs_front_fwd leaves behind, with fewer and smaller passes:
//   * the depth key is 24 bits wide when the engine knows the view's depth range (key = depth bits - key_base, checked by the front
//     kernel): THREE V-sized passes instead of four;
//   * the tile rectangles come packed from the front kernel (no tile_rect_kernel);
//   * the emission finds its own write positions with an in-launch chained scan over its blocks (ticket order, 8-byte {flag, value}
//     granules as in project_fwd_kernel): no block-sum pass, no scan launch;
//   * the front kernel has already counted the intersections per TILE (LDS histogram per block of 512 index-adjacent Gaussians): the
//     tile offsets are the exclusive scan of those 2 500 counters, clamped to the capacity -- no I-sized offsets pass, and the last
//     tile pass writes 4 bytes per intersection (flatten id) instead of 8.  (More than 8 192 tiles: the tile ids are kept and the
//     offsets come from them.)
// 18 launches (setup, 3 x 3 depth passes, emission, offset scan, 2 x 3 tile passes) against 25.
// Round 6, built, bit-identical, measured and removed (`git show 3d98652:geosplatting_amd/csrc/gs_sort.hip`): every pass as ONE kernel
// -- ticketed blocks publish {flag | count} rows, offsets from a two-level look-back over <= 31 block rows + the group totals, items held
// in registers meanwhile, global digit histograms from one read of the keys / from the tile counters, the tile rectangle carried through
// the depth passes as 12-byte items so that the emission reads in order (48 instead of 84 us) -- 9 launches instead of 18.  A fused pass
// took 36 us (depth) / 53 us (tiles) alone against 8 + 6 + 16 / 8 + 6 + 18 for the three kernels it replaces: with every block resident
// and publishing at the same moment the look-back is ~10 dependent global round trips, where a kernel boundary costs 1.5 us; 281 against
// 252 us per view alone, 640 against 649 views/s inside the step (608 with 32 loads in flight and 21 KB blocks).
struct KeyIn {                                                     // (depth key, packed index) straight from the key array
    const unsigned* keys;
    __device__ __forceinline__ uint2 load(int64_t i) const { return make_uint2(keys[i], (unsigned)i); }
};
struct FinalFlat {
    int32_t* flatten_ids;
    __device__ __forceinline__ void store(int64_t i, const uint2& it) const { flatten_ids[i] = (int32_t)it.y; }
};

#define BF_HIST_MAX 8192                     // tiles whose counters fit a block's LDS histogram (32 KB)

__global__ void __launch_bounds__(256)
bin_setup_kernel(unsigned* __restrict__ state, int n_state_words,
                 const long long* __restrict__ counts, long long v_cap, long long i_cap, long long* __restrict__ status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (int k = i; k < n_state_words; k += stride) state[k] = 0u;
    if (i == 0 && counts != nullptr && status != nullptr) {                        // capacity protocol, as capacity_check_kernel
        const long long V = counts[0], I = counts[1];
        if (V > v_cap || I > i_cap) {
            status[0] = GS_ENOSPC;
            if (I > status[1]) status[1] = I;
            if (V > status[2]) status[2] = V;
        }
    }
}

#define BF_VALID (1ull << 63)
#define BF_SPIN_LIMIT (1 << 22)

// thread r (depth rank) writes the (tile, index) items of its Gaussian at the exclusive prefix of the tile counts in depth order; the
// prefix across blocks is a decoupled look-back (blocks numbered by ticket, so a block only waits for blocks that already run)
__global__ void __launch_bounds__(EM_THREADS)
emit_chained_kernel(GsCount vc, const uint2* __restrict__ order, const uint2* __restrict__ rect, unsigned* __restrict__ ctrl,
                    u64* __restrict__ desc, int n_blocks, int tile_w, uint2* __restrict__ items, unsigned item_cap,
                    long long* __restrict__ status /* nullable: capacity protocol word */)
{
    const int V = (int)gs_count(vc);
    __shared__ unsigned ws[EM_THREADS / 64];
    __shared__ unsigned s_block;
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_block = atomicAdd(ctrl, 1u);
    __syncthreads();
    const int block = (int)s_block;
    u64* agg = desc; u64* pre = desc + n_blocks;
    // blocked arrangement: thread t owns ranks base + t*EM_PER .. +EM_PER-1 (consecutive: the scan stays in index order)
    const int r0 = block * EM_TILE + (int)threadIdx.x * EM_PER;
    int v[EM_PER]; unsigned c[EM_PER]; uint2 q[EM_PER];
    unsigned mine = 0u;
#pragma unroll
    for (int k = 0; k < EM_PER; ++k) {
        const int r = r0 + k;
        v[k] = r < V ? (int)order[r].y : -1;
        q[k] = v[k] >= 0 ? rect[v[k]] : make_uint2(0u, 0u);
        c[k] = rect_count(q[k]);
        mine += c[k];
    }
    unsigned incl = mine;
    incl = gs_wave_incl_scan(incl);
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    unsigned before = 0u, total = 0u;
    for (int w = 0; w < EM_THREADS / 64; ++w) { if (w < wave) before += ws[w]; total += ws[w]; }
    if (wave == 0) {
        // look-back over EM_LOOK x 64 predecessors per round, nearest first: every lane has EM_LOOK descriptor pairs in flight and
        // keeps a private partial sum; ONE wave reduction at the end.  (All 1 920 blocks of a 2 M-Gaussian view are resident at once
        // and publish their aggregates together: nobody has a prefix yet, and block b walks all the way back -- b / 64 rounds of
        // dependent L2 round trips with a 64-bit wave sum each was most of this kernel's 84 us, round 6.)
        unsigned long long acc = 0ull;
        if (block > 0) {
            if (lane == 0) __hip_atomic_store(&agg[block], (u64)total | BF_VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int pos = block - 1;
            bool done = false;
            while (!done) {
                u64 pv[EM_LOOK], av[EM_LOOK];
#pragma unroll
                for (int j = 0; j < EM_LOOK; ++j) {
                    const int idx = pos - 64 * j - lane;
                    pv[j] = av[j] = 0ull;
                    if (idx >= 0) {
                        pv[j] = __hip_atomic_load(&pre[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        av[j] = __hip_atomic_load(&agg[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int j = 0; j < EM_LOOK; ++j) {
                    if (done) break;                         // (wave-uniform: the groups behind the first prefix are neither awaited nor summed)
                    const int idx = pos - 64 * j - lane;
                    bool isP = idx < 0;                      // (virtual predecessors before block 0: prefix 0)
                    u64 val = 0ull;
                    if (idx >= 0) {
                        int spins = 0;
                        for (;;) {
                            if (pv[j] & BF_VALID) { isP = true; val = pv[j]; break; }
                            if (av[j] & BF_VALID) { val = av[j]; break; }
                            if (++spins > BF_SPIN_LIMIT) { atomicExch(ctrl + 1, 1u); if (status) status[0] = GS_ENOSPC; break; }   // (reported as a truncated view)
                            __builtin_amdgcn_s_sleep(2);
                            pv[j] = __hip_atomic_load(&pre[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            av[j] = __hip_atomic_load(&agg[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    const u64 pmask = __ballot(isP);
                    const int first = pmask ? __builtin_ctzll(pmask) : 64;
                    if (lane <= first) acc += val & ~BF_VALID;
                    if (pmask) done = true;
                }
                pos -= 64 * EM_LOOK;
            }
        }
        const unsigned long long base = gs_wave_sum_u64(acc);
        if (lane == 0) {
            __hip_atomic_store(&pre[block], (u64)(base + total) | BF_VALID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base = base;
        }
    }
    __syncthreads();
    unsigned long long cur = s_base + before + incl - mine;
#pragma unroll
    for (int k = 0; k < EM_PER; ++k) {
        if (v[k] < 0 || c[k] == 0u) continue;
        const int x0 = (int)(q[k].x & 0xffffu), y0 = (int)(q[k].x >> 16), x1 = (int)(q[k].y & 0xffffu), y1 = (int)(q[k].y >> 16);
        for (int i = y0; i < y1; ++i)
            for (int j = x0; j < x1; ++j) {
                const unsigned t = (unsigned)(i * tile_w + j);
                if (cur < (unsigned long long)item_cap) items[cur] = make_uint2(t, (unsigned)v[k]);   // (capacity protocol: an overflowing view is reported, not written)
                ++cur;
            }
    }
}

// offsets[t] = min(exclusive prefix of the tile counts, n) by one workgroup: the tile offsets of a view.  (The clamp only ever acts
// on a view that overflowed its capacity: its list is truncated in emission order, the offsets then are not those of the truncated
// list -- memory-safe, wrong image, reported, as the capacity protocol says.)
__global__ void __launch_bounds__(1024)
tile_offsets_scan_kernel(int n_tiles, const unsigned* __restrict__ tile_counts, GsCount nc, int32_t* __restrict__ offsets)
{
    __shared__ unsigned wsum[16];
    const unsigned n = (unsigned)gs_count(nc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned carry = 0u;
    for (int i0 = 0; i0 < n_tiles; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const unsigned v = i < n_tiles ? tile_counts[i] : 0u;
        unsigned incl = v;
        incl = gs_wave_incl_scan(incl);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned before = 0u, all = 0u;
        for (int w = 0; w < 16; ++w) { if (w < wave) before += wsum[w]; all += wsum[w]; }
        if (i < n_tiles) { const unsigned o = carry + before + incl - v; offsets[i] = (int32_t)(o < n ? o : n); }
        carry += all;
        __syncthreads();
    }
}

// offsets[t] = first sorted position whose tile id >= t (the > 8 192-tile path; same as isect_offsets_tiles_kernel of gs_project.hip)
__global__ void __launch_bounds__(256)
bin_offsets_tiles_kernel(GsCount nc, const int32_t* __restrict__ tiles, int n_tiles, int32_t* __restrict__ offsets)
{
    const int64_t n = gs_count(nc);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0 && i == 0)
        for (int t = 0; t < n_tiles; ++t) offsets[t] = 0;
    if (i >= n) return;
    const int cur = tiles[i];
    if (i == 0) {
        for (int t = 0; t <= cur && t < n_tiles; ++t) offsets[t] = 0;
    } else {
        const int prev = tiles[i - 1];
        for (int t = prev + 1; t <= cur && t < n_tiles; ++t) offsets[t] = (int32_t)i;
    }
    if (i == n - 1)
        for (int t = cur + 1; t < n_tiles; ++t) offsets[t] = (int32_t)n;
}


extern "C" size_t gs_isect_bin_front_ws_bytes(int V, int64_t n_isects, int tile_w, int tile_h)
{
    const size_t v = V > 0 ? (size_t)V : 1, n = n_isects > 0 ? (size_t)n_isects : 1;
    const size_t tb = table_bytes((int64_t)(v > n ? v : n));
    return tb + 2 * align256(v * 8) + 2 * align256(n * 8) + bf_state_bytes((int)v) + align256(n * 4) + 256;   // (tile ids: only without tile_counts)
}

extern "C" int gs_isect_bin_front(int V_cap, const uint32_t* depth_keys, const uint32_t* tile_rects, const uint32_t* tile_counts,
                                  const int64_t* counts_dev, int64_t n_isects_cap, int key_bits, int tile_w, int tile_h, int32_t* flatten_ids_sorted,
                                  int32_t* isect_offsets, void* ws, size_t ws_bytes, int64_t* status_dev, void* stream)
{
    const int V = V_cap;
    const int64_t n_isects = n_isects_cap;
    GS_CHECK_ARG(V >= 0 && n_isects >= 0 && tile_w > 0 && tile_h > 0 && (key_bits == 24 || key_bits == 32), "bad sizes or key width");
    GS_CHECK_ARG(n_isects < (1ll << 31), "n_isects must fit int32");
    GS_CHECK_ARG((int64_t)tile_w * tile_h < (1ll << 24) && tile_w < 65536 && tile_h < 65536, "tile grid too large");
    GS_CHECK_ARG(isect_offsets != nullptr && (counts_dev == nullptr || status_dev != nullptr), "null argument");
    hipStream_t s = (hipStream_t)stream;
    const int n_tiles = tile_w * tile_h;
    if (V == 0 || n_isects == 0) { GS_CHECK_HIP(gs_zero_async(isect_offsets, sizeof(int32_t) * (size_t)n_tiles, s)); return GS_OK; }
    GS_CHECK_ARG(depth_keys != nullptr && tile_rects != nullptr && flatten_ids_sorted != nullptr && ws != nullptr, "null argument");
    if (ws_bytes < gs_isect_bin_front_ws_bytes(V, n_isects, tile_w, tile_h)) { gs_set_error("gs_isect_bin_front: workspace too small"); return GS_ENOSPC; }
    const GsCount vc{ (long long)V, (const long long*)counts_dev }, ic{ (long long)n_isects, counts_dev ? (const long long*)counts_dev + 1 : nullptr };
    const bool hist = tile_counts != nullptr && n_tiles <= BF_HIST_MAX;     // otherwise the offsets come from the sorted tile ids
    char* p = (char*)ws;
    unsigned* table = (unsigned*)p; p += table_bytes((int64_t)V > n_isects ? (int64_t)V : n_isects);
    uint2* da = (uint2*)p; p += align256((size_t)V * 8);
    uint2* db = (uint2*)p; p += align256((size_t)V * 8);
    uint2* ia = (uint2*)p; p += align256((size_t)n_isects * 8);
    uint2* ib = (uint2*)p; p += align256((size_t)n_isects * 8);
    unsigned* state = (unsigned*)p; p += bf_state_bytes(V);
    int32_t* tile_ids = hist ? nullptr : (int32_t*)p;
    const int eblocks = (V + EM_TILE - 1) / EM_TILE;
    const int n_state_words = (int)(bf_state_bytes(V) / 4);
    // 0. clear the emission's look-back state and the tile counters; capacity check
    hipLaunchKernelGGL(bin_setup_kernel, dim3(gs_cdiv(n_state_words, 256) < 64 ? gs_cdiv(n_state_words, 256) : 64),
                       dim3(256), 0, s, state, n_state_words, (const long long*)counts_dev,
                       (long long)V_cap, (long long)n_isects_cap, (long long*)status_dev);
    GS_CHECK_LAUNCH();
    // 1. depth order of the Gaussians: key_bits / 8 stable 8-bit passes over (key, index); the first reads the key array
    int rc = radix_pass<uint2>(vc, KeyIn{ depth_keys }, XDigit{ 0, 255u }, U2Out{ da }, 8, table, s);
    if (rc != GS_OK) return rc;
    rc = radix_pass<uint2>(vc, U2In{ da }, XDigit{ 8, 255u }, U2Out{ db }, 8, table, s);
    if (rc != GS_OK) return rc;
    rc = radix_pass<uint2>(vc, U2In{ db }, XDigit{ 16, 255u }, U2Out{ da }, 8, table, s);
    if (rc != GS_OK) return rc;
    const uint2* order = da;
    if (key_bits == 32) {
        rc = radix_pass<uint2>(vc, U2In{ da }, XDigit{ 24, 255u }, U2Out{ db }, 8, table, s);
        if (rc != GS_OK) return rc;
        order = db;
    }
    // 2. emission in depth order (chained scan inside the launch) + per-tile counts
    unsigned* ctrl = state; u64* desc = (u64*)((char*)state + 16);
    hipLaunchKernelGGL(emit_chained_kernel, dim3(eblocks), dim3(EM_THREADS), 0, s, vc, order, (const uint2*)tile_rects, ctrl, desc,
                       eblocks + 1, tile_w, ia, (unsigned)n_isects, (long long*)status_dev);
    GS_CHECK_LAUNCH();
    if (hist) {
        hipLaunchKernelGGL(tile_offsets_scan_kernel, dim3(1), dim3(1024), 0, s, n_tiles, (const unsigned*)tile_counts, ic, isect_offsets);
        GS_CHECK_LAUNCH();
    }
    // 3. stable split by tile id: npass digits of `width` bits, the last one writes the sorted flatten ids
    int tb = 0;
    while ((1 << tb) < n_tiles) ++tb;
    if (tb < 1) tb = 1;
    const int npass = (tb + 7) / 8, width = (tb + npass - 1) / npass;
    uint2* src = ia; uint2* dst = ib;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = pass * width;
        const int nbits = (tb - shift) < width ? (tb - shift) : width;
        if (pass == npass - 1 && hist)
            rc = radix_pass<uint2>(ic, U2In{ src }, XDigit{ shift, (1u << nbits) - 1u }, FinalFlat{ flatten_ids_sorted }, nbits, table, s);
        else if (pass == npass - 1)
            rc = radix_pass<uint2>(ic, U2In{ src }, XDigit{ shift, (1u << nbits) - 1u }, FinalTiles{ tile_ids, flatten_ids_sorted }, nbits, table, s);
        else
            rc = radix_pass<uint2>(ic, U2In{ src }, XDigit{ shift, (1u << nbits) - 1u }, U2Out{ dst }, nbits, table, s);
        if (rc != GS_OK) return rc;
        uint2* t = src; src = dst; dst = t;
    }
    if (!hist) {
        hipLaunchKernelGGL(bin_offsets_tiles_kernel, dim3(gs_cdiv(n_isects, 256)), dim3(256), 0, s, ic, tile_ids, n_tiles, isect_offsets);
        GS_CHECK_LAUNCH();
    }
    return GS_OK;
}

// ---- gs_isect_bin / _cap / _tiles_cap: the rasterization() call shape (emitted keys are never materialised) -----------------------
static int isect_bin_impl(int V, const float* means2d, const int32_t* radii, const float* depths, int64_t n_isects,
                          const long long* counts_dev, int tile_size, int tile_w, int tile_h, int64_t* isect_ids_sorted,
                          int32_t* flatten_ids_sorted, void* ws, size_t ws_bytes, void* stream, int32_t* tile_ids_sorted)
{
    const GsCount vc{ (long long)V, counts_dev }, ic{ (long long)n_isects, counts_dev ? counts_dev + 1 : nullptr };
    GS_CHECK_ARG(V >= 0 && n_isects >= 0 && tile_size > 0 && tile_w > 0 && tile_h > 0, "bad sizes");
    GS_CHECK_ARG(n_isects < (1ll << 31), "n_isects must fit int32");
    GS_CHECK_ARG((int64_t)tile_w * tile_h < (1ll << 24), "more than 2^24 tiles");
    if (V == 0 || n_isects == 0) return GS_OK;
    GS_CHECK_ARG(ws != nullptr, "workspace must not be NULL");
    if (ws_bytes < gs_isect_bin_ws_bytes(V, n_isects, tile_w, tile_h)) { gs_set_error("gs_isect_bin: workspace too small"); return GS_ENOSPC; }
    hipStream_t s = (hipStream_t)stream;
    char* p = (char*)ws;
    unsigned* table = (unsigned*)p; p += table_bytes((int64_t)V > n_isects ? (int64_t)V : n_isects);
    uint2* da = (uint2*)p; p += align256((size_t)V * 8);
    uint2* db = (uint2*)p; p += align256((size_t)V * 8);
    uint2* rect = (uint2*)p; p += align256((size_t)V * 8);
    const int eblocks = (V + EM_TILE - 1) / EM_TILE;
    unsigned* state = (unsigned*)p; p += bf_state_bytes(V);     // look-back state of the emission
    uint2* ia = (uint2*)p; p += align256((size_t)n_isects * 8);
    uint2* ib = (uint2*)p; p += align256((size_t)n_isects * 8);
    int tb = 0;
    while ((1 << tb) < tile_w * tile_h) ++tb;
    if (tb < 1) tb = 1;
    const int npass = (tb + 7) / 8, width = (tb + npass - 1) / npass;
    int rc = GS_OK;
    // 1. depth order of the Gaussians: four stable 8-bit passes over (depth bits, index); the first reads the depth array
    rc = radix_pass<uint2>(vc, DepthIn{ depths }, XDigit{ 0, 255u }, U2Out{ da }, 8, table, s);
    if (rc != GS_OK) return rc;
    rc = radix_pass<uint2>(vc, U2In{ da }, XDigit{ 8, 255u }, U2Out{ db }, 8, table, s);
    if (rc != GS_OK) return rc;
    rc = radix_pass<uint2>(vc, U2In{ db }, XDigit{ 16, 255u }, U2Out{ da }, 8, table, s);
    if (rc != GS_OK) return rc;
    rc = radix_pass<uint2>(vc, U2In{ da }, XDigit{ 24, 255u }, U2Out{ db }, 8, table, s);
    if (rc != GS_OK) return rc;
    // 2. emission in depth order: positions from a chained scan inside the launch (emit_chained_kernel, as gs_isect_bin_front)
    GS_CHECK_ARG(tile_w < 65536 && tile_h < 65536, "tile grid too large");
    hipLaunchKernelGGL(tile_rect_kernel, dim3(gs_cdiv(V, 256)), dim3(256), 0, s, vc, means2d, radii, tile_size, tile_w, tile_h, rect);
    GS_CHECK_LAUNCH();
    {
        const int n_state_words = (int)(bf_state_bytes(V) / 4);
        hipLaunchKernelGGL(bin_setup_kernel, dim3(gs_cdiv(n_state_words, 256) < 64 ? gs_cdiv(n_state_words, 256) : 64), dim3(256), 0, s, state,
                           n_state_words, (const long long*)nullptr, 0ll, 0ll, (long long*)nullptr);
        GS_CHECK_LAUNCH();
        hipLaunchKernelGGL(emit_chained_kernel, dim3(eblocks), dim3(EM_THREADS), 0, s, vc, (const uint2*)db, (const uint2*)rect, state,
                           (u64*)((char*)state + 16), eblocks + 1, tile_w, ia, (unsigned)n_isects, (long long*)nullptr);
        GS_CHECK_LAUNCH();
    }
    // 3. stable split by tile id: npass digits of `width` bits, the last one writes the sorted meta arrays
    uint2* src = ia; uint2* dst = ib;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = pass * width;
        const int nbits = (tb - shift) < width ? (tb - shift) : width;
        if (pass == npass - 1 && tile_ids_sorted != nullptr)
            rc = radix_pass<uint2>(ic, U2In{ src }, XDigit{ shift, (1u << nbits) - 1u },
                                   FinalTiles{ tile_ids_sorted, flatten_ids_sorted }, nbits, table, s);
        else if (pass == npass - 1)
            rc = radix_pass<uint2>(ic, U2In{ src }, XDigit{ shift, (1u << nbits) - 1u },
                                   FinalOut{ (u64*)isect_ids_sorted, flatten_ids_sorted, depths }, nbits, table, s);
        else
            rc = radix_pass<uint2>(ic, U2In{ src }, XDigit{ shift, (1u << nbits) - 1u }, U2Out{ dst }, nbits, table, s);
        if (rc != GS_OK) return rc;
        uint2* t = src; src = dst; dst = t;
    }
    return GS_OK;
}
