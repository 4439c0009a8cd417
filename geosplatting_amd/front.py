"""The engine's fused per-view front and tail (csrc/gs_front.hip, gs_isect_bin_front in csrc/gs_sort.hip), as Python stages.

Reference cut: ``RenderableAttrs.splat`` shades every Gaussian (rfstudio/model/geosplat.py:80-122) and hands the colours to
``gsplat.rasterization`` (rfstudio/model/gsplat.py:334-355); autograd then runs the projection backward and the shading backward
one after the other.  ``geosplatting_amd.shade`` + ``geosplatting_amd.rasterization`` keep that call shape.  The step engine does
not need it: ``front_stage`` is ONE launch (projection with the shading fused ahead of it, north star), ``bin_stage`` the binning
from what it leaves behind, ``tail_stage`` ONE launch for the projection + shading backward.  Per visible Gaussian the path holds
one 64-byte record, a 4-byte depth key and an 8-byte tile rectangle -- nothing else.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from . import _lib as L

_pinned_pool4 = []           # pinned int64[4] buffers for the asynchronous {V, I, ~min depth bits, max depth bits} read-back


def pinned_counts4() -> Tensor:
    return _pinned_pool4.pop() if _pinned_pool4 else torch.empty(4, dtype=torch.int64).pin_memory()


def release_counts4(t: Tensor) -> None:
    _pinned_pool4.append(t)


def reserve_pinned(n: int) -> None:
    """Pinned buffers cannot be allocated while a stream is capturing: make sure `n` are waiting."""
    while len(_pinned_pool4) < n:
        _pinned_pool4.append(torch.empty(4, dtype=torch.int64).pin_memory())


class Front:
    """What gs_front_fwd left behind for one view; (V, I, depth range) on their way to pinned memory."""
    __slots__ = ("vis", "keys", "rects", "tile_counts", "packed_index", "counts", "host_counts", "event", "whs", "key_bits", "N")


def front_stage(means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, normals: Tensor, kd: Tensor, ks: Tensor,
                viewmat: Tensor, K: Tensor, cam_pos: Tensor, env_struct, W: int, H: int, min_roughness: float, max_metallic: float,
                mode: int, key_base: int = 0, key_bits: int = 32, status: Optional[Tensor] = None, tile_size: int = 16,
                eps2d: float = 0.3, near: float = 0.01, far: float = 1e10, radius_clip: float = 0.0, want_packed_index: bool = False,
                records: bool = True, binning: bool = True, tight_tiles: bool = False) -> Front:
    """S1-S3 + A1 (+ A1', A2 count) of one view on the current stream.  `scales` / `opacities` are the activated values
    (rfstudio/model/gsplat.py:336-339).  key_bits 24 needs `status` (int64[4]; word 3 reports a depth outside the key range).
    records=False: geometry only (keys, rectangles, tile counts; env_struct may be None); binning=False: records only.
    tight_tiles: tile rectangles clipped to the {alpha >= 1/255} extents (fewer intersections, identical pixels; not gsplat's `meta`)."""
    lib = L.lib()
    dev = means.device
    N = means.shape[0]
    fr = Front()
    fr.vis = torch.empty(max(N, 1), 16, dtype=torch.float32, device=dev) if records else None
    fr.keys = torch.empty(max(N, 1), dtype=torch.int32, device=dev) if binning else None
    fr.rects = torch.empty(max(N, 1), 2, dtype=torch.int32, device=dev) if binning else None
    fr.counts = torch.empty(4, dtype=torch.int64, device=dev)
    tw, th = (W + tile_size - 1) // tile_size, (H + tile_size - 1) // tile_size
    # (above 8 192 tiles the front kernel keeps no per-tile histogram and the binning derives the offsets from the sorted tile ids)
    fr.tile_counts = torch.empty(tw * th, dtype=torch.int32, device=dev) if (binning and tw * th <= 8192) else None
    fr.packed_index = torch.empty(max(N, 1), dtype=torch.int32, device=dev) if want_packed_index else None
    ws_bytes = lib.gs_front_ws_bytes(N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    L.check(lib.gs_front_fwd(N, L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(opacities), L.ptr(normals), L.ptr(kd), L.ptr(ks),
                             L.ptr(viewmat), L.ptr(K), L.ptr(cam_pos), L.f32(min_roughness), L.f32(max_metallic), mode,
                             C.byref(env_struct) if env_struct is not None else None, W, H, tile_size, L.f32(eps2d), L.f32(near), L.f32(far), L.f32(radius_clip),
                             C.c_uint32(int(key_base) & 0xffffffff), int(key_bits), L.ptr(fr.vis), L.ptr(fr.keys), L.ptr(fr.rects),
                             L.ptr(fr.tile_counts), L.ptr(fr.packed_index), 1 if tight_tiles else 0, L.ptr(fr.counts), L.ptr(status), L.ptr(ws), C.c_size_t(ws_bytes), L.stream()), "gs_front_fwd")
    fr.host_counts = pinned_counts4()
    fr.host_counts.copy_(fr.counts, non_blocking=True)        # 32 bytes, asynchronous
    fr.event = torch.cuda.Event()
    fr.event.record()
    fr.whs = (W, H, tile_size)
    fr.key_bits = int(key_bits)
    fr.N = N
    return fr


def depth_range(host_counts: Tensor):
    """(min, max) depth bits of a finished view from its pinned counts, or None when nothing was visible."""
    if int(host_counts[0]) <= 0:
        return None
    return 0xffffffff - int(host_counts[2]), int(host_counts[3])


def bin_stage(fr: Front, i_cap: Optional[int], status: Optional[Tensor], prepare: bool = True, binned=None):
    """A2-A4 + the compositor's record stream for a front; exact mode (i_cap None) waits for that view's (V, I) -- the one host
    synchronisation of the forward, as upstream; capacity mode sizes everything by (N, i_cap) and reads the counts on the device.
    Returns (state, V, I) with V / I the sizes the later launches are given (capacities in capacity mode).
    prepare=False: binning only (a geometry-only front); binned=(flatten_ids, isect_offsets) of such an earlier call: record stream only."""
    lib = L.lib()
    W, H, tile_size = fr.whs
    dev = fr.counts.device
    tw, th = (W + tile_size - 1) // tile_size, (H + tile_size - 1) // tile_size
    st = L.stream()
    if i_cap is None:
        fr.event.synchronize()
        V, I = int(fr.host_counts[0]), int(fr.host_counts[1])
        if V < 0 or I < 0 or I >= 2 ** 31:
            raise L.GeoSplatHipError(f"bad intersection count V={V} I={I}")
        counts = None
    else:
        V, I, counts = fr.N, int(i_cap), fr.counts
    if binned is not None:
        flat, offsets = binned
    else:
        flat = torch.empty(max(I, 1), dtype=torch.int32, device=dev)
        offsets = torch.empty(th * tw, dtype=torch.int32, device=dev)
        bin_bytes = lib.gs_isect_bin_front_ws_bytes(V, L.i64(I), tw, th)
        bin_ws = torch.empty(max(bin_bytes, 1), dtype=torch.uint8, device=dev)
        L.check(lib.gs_isect_bin_front(V, L.ptr(fr.keys), L.ptr(fr.rects), L.ptr(fr.tile_counts), L.ptr(counts), L.i64(I), fr.key_bits, tw, th,
                                       L.ptr(flat), L.ptr(offsets), L.ptr(bin_ws), C.c_size_t(bin_bytes),
                                       L.ptr(status) if counts is not None else None, st), "gs_isect_bin_front")
    if not prepare:
        return dict(flatten_ids=flat, isect_offsets=offsets, counts=counts, keys=fr.keys, rects=fr.rects, tile_counts=fr.tile_counts), V, I
    rws_bytes = lib.gs_raster_ws_bytes(L.i64(I), V, W, H, tile_size)
    rws = torch.empty(rws_bytes, dtype=torch.uint8, device=dev)
    if counts is None:
        L.check(lib.gs_raster_prepare_vis(W, H, tile_size, 3, V, L.ptr(fr.vis), L.i64(I), L.ptr(offsets), L.ptr(flat), L.ptr(rws),
                                          C.c_size_t(rws_bytes), st), "gs_raster_prepare_vis")
    else:
        L.check(lib.gs_raster_prepare_vis_cap(W, H, tile_size, 3, V, L.ptr(fr.vis), L.i64(I), L.ptr(counts), L.ptr(offsets), L.ptr(flat),
                                              L.ptr(rws), C.c_size_t(rws_bytes), st), "gs_raster_prepare_vis_cap")
    state = dict(vis_records=fr.vis, flatten_ids=flat, isect_offsets=offsets, raster_ws=rws, counts=counts, keys=fr.keys, rects=fr.rects,
                 tile_counts=fr.tile_counts, packed_index=fr.packed_index)
    return state, V, I


def tail_stage(V: int, counts: Optional[Tensor], means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, normals: Tensor,
               kd: Tensor, ks: Tensor, viewmat: Tensor, K: Tensor, cam_pos: Tensor, env_struct, env_grad_struct, W: int, H: int,
               min_roughness: float, max_metallic: float, mode: int, vis: Tensor, v_packed: Tensor, g_means: Tensor, g_quats: Tensor,
               g_scales: Tensor, g_opacities: Tensor, g_normals: Tensor, g_kd: Tensor, g_ks: Tensor, eps2d: float = 0.3,
               priv: Optional[Tensor] = None) -> None:
    """A7 + S1-S3 backward of one view on the current stream: ADDS into the gradient buffers and the texel gradients.
    `priv`: the zeroed scratch of `tail_priv_alloc` (XCD-private copies of the mid-sized levels; fold with `tail_priv_reduce`)."""
    lib = L.lib()
    L.check(lib.gs_tail_bwd(V, L.ptr(counts), L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(opacities), L.ptr(normals), L.ptr(kd),
                            L.ptr(ks), L.ptr(viewmat), L.ptr(K), L.ptr(cam_pos), L.f32(min_roughness), L.f32(max_metallic), mode,
                            C.byref(env_struct), W, H, L.f32(eps2d), L.ptr(vis), L.ptr(v_packed), int(v_packed.shape[1]),
                            L.ptr(g_means), L.ptr(g_quats), L.ptr(g_scales), L.ptr(g_opacities), L.ptr(g_normals), L.ptr(g_kd),
                            L.ptr(g_ks), C.byref(env_grad_struct), L.ptr(priv), C.c_size_t(0 if priv is None else priv.numel()),
                            L.stream()), "gs_tail_bwd")


def tail_priv_alloc(env_struct, mode: int, device) -> Optional[Tensor]:
    """Zeroed scratch for the XCD-private texel-gradient copies of one step (None when the pyramid has no mid-sized level)."""
    n = L.lib().gs_tail_priv_ws_bytes(C.byref(env_struct), mode)
    return torch.zeros(n, dtype=torch.uint8, device=device) if n > 0 else None


def tail_priv_reduce(env_struct, env_grad_struct, mode: int, priv: Optional[Tensor]) -> None:
    if priv is not None:
        L.check(L.lib().gs_tail_priv_reduce(C.byref(env_struct), mode, L.ptr(priv), C.c_size_t(priv.numel()), C.byref(env_grad_struct),
                                            L.stream()), "gs_tail_priv_reduce")


def tail_multi_stage(views, means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, normals: Tensor, kd: Tensor, ks: Tensor,
                     env_struct, env_grad_struct, min_roughness: float, max_metallic: float, mode: int, g_means: Tensor, g_quats: Tensor,
                     g_scales: Tensor, g_opacities: Tensor, g_normals: Tensor, g_kd: Tensor, g_ks: Tensor, accumulate: bool,
                     eps2d: float = 0.3, priv: Optional[Tensor] = None, parts: int = 3) -> None:
    """A7 + S1-S3 backward of SEVERAL views in one call (gs_tail_bwd_multi_parts): `views` = list of
    (viewmat, K, cam_pos, vis_records, v_packed, packed_index, W, H).  accumulate False: the gradient buffers are overwritten.
    parts: 1 = the shading half only, 2 = the projection half only (after the shading half of the same call), 3 = both;
    + 4 = a background launch (the shading half on half of the CUs, beside the compositor of the following views)."""
    lib = L.lib()
    arr = (L.GsTailView * len(views))()
    stride = None
    for i, (vm, K, cp, vis, vp, pidx, W, H) in enumerate(views):
        arr[i].viewmat = vm.data_ptr(); arr[i].K = K.data_ptr(); arr[i].cam_pos = cp.data_ptr()
        arr[i].vis_records = vis.data_ptr(); arr[i].v_packed = vp.data_ptr(); arr[i].packed_index = pidx.data_ptr()
        arr[i].W = int(W); arr[i].H = int(H)
        assert stride in (None, int(vp.shape[1]))
        stride = int(vp.shape[1])
    L.check(lib.gs_tail_bwd_multi_parts(int(parts), means.shape[0], len(views), arr, L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(opacities), L.ptr(normals),
                                  L.ptr(kd), L.ptr(ks), L.f32(min_roughness), L.f32(max_metallic), mode, C.byref(env_struct), L.f32(eps2d),
                                  stride, L.ptr(g_means), L.ptr(g_quats), L.ptr(g_scales), L.ptr(g_opacities), L.ptr(g_normals), L.ptr(g_kd),
                                  L.ptr(g_ks), 1 if accumulate else 0, C.byref(env_grad_struct), L.ptr(priv),
                                  C.c_size_t(0 if priv is None else priv.numel()), L.stream()), "gs_tail_bwd_multi_parts")
