// gs_sort.hip -- A3: stable ascending sort of the (tile|depth) keys with their packed-index payload.
// Semantics: cub::DeviceRadixSort::SortPairs over the low 32 + tile_bits key bits (gsplat 1.4
// `isect_tiles(sort=True)`, reached from rfstudio/model/gsplat.py:334).  Stability makes equal keys keep
// their emission order (= ascending packed index), which is what fixes the composited order bit-exactly.
//
// Round-1 implementation: rocPRIM's device-wide LSD radix sort (header-only, compiled for gfx950 into this
// library) restricted to the significant bits.  It is HBM-bound: ~(12 B read + 12 B write) per
// intersection per 8-bit digit pass.  DESIGN.md lists the planned replacement (tile-binned scatter +
// per-tile LDS bitonic with index tie-break) that touches each key once.
#include "gs_common.h"

#include <cstring>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>

static int tile_bits(int tile_w, int tile_h)
{
    // floor(log2(n_tiles)) + 1
    unsigned n = (unsigned)(tile_w * tile_h);
    int b = 0;
    while (n > 1) { n >>= 1; ++b; }
    return b + 1;
}

extern "C" size_t gs_sort_ws_bytes(int64_t n_isects, int tile_w, int tile_h)
{
    if (n_isects <= 0) return 0;
    size_t bytes = 0;
    const int end_bit = 32 + tile_bits(tile_w, tile_h);
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                             (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)n_isects, 0,
                                             (unsigned)end_bit, (hipStream_t)0);
    if (e != hipSuccess) return 0;
    return bytes + 256;
}

extern "C" int gs_isect_sort(int64_t n_isects, const int64_t* isect_ids, const int32_t* flatten_ids,
                             int64_t* isect_ids_sorted, int32_t* flatten_ids_sorted, int tile_w, int tile_h,
                             void* ws, size_t ws_bytes, void* stream)
{
    GS_CHECK_ARG(n_isects >= 0 && tile_w > 0 && tile_h > 0, "bad sizes");
    if (n_isects == 0) return GS_OK;
    size_t need = 0;
    const int end_bit = 32 + tile_bits(tile_w, tile_h);
    GS_CHECK_HIP(rocprim::radix_sort_pairs(nullptr, need, (const uint64_t*)isect_ids, (uint64_t*)isect_ids_sorted,
                                           flatten_ids, flatten_ids_sorted, (size_t)n_isects, 0, (unsigned)end_bit,
                                           (hipStream_t)stream));
    if (ws_bytes < need) { gs_set_error("gs_isect_sort: workspace too small (%zu < %zu)", ws_bytes, need); return GS_ENOSPC; }
    GS_CHECK_HIP(rocprim::radix_sort_pairs(ws, need, (const uint64_t*)isect_ids, (uint64_t*)isect_ids_sorted,
                                           flatten_ids, flatten_ids_sorted, (size_t)n_isects, 0, (unsigned)end_bit,
                                           (hipStream_t)stream));
    return GS_OK;
}
