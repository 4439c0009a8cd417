"""Drop-in for the rasterizer boundary of the reference: ``gsplat.rasterization`` exactly as it is called at
rfstudio/model/gsplat.py:334-355 (and :240-261; rfstudio/model/geosplat.py:276-295 with D=14).

Same argument names / meaning / return triple ``(render[1,H,W,D], alpha[1,H,W,1], meta)``; differentiable
w.r.t. means / quats / scales / opacities / colors.  Every stage runs in libgeosplat_hip.so (hand-written HIP
for gfx950) through the C-ABI in include/geosplat_hip.h -- there is no PyTorch or CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L

_META_KEYS = ("gaussian_ids_i32", "radii", "means2d", "depths", "conics", "compensations", "opacities", "colors",
              "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets", "last_ids", "raster_ws")


class _Projected:
    """Outputs of the projection stage (A1 + A1' + A2 count) of one view, with the (V, I) counts on their way to a
    pinned host buffer.  Nothing here blocks: `counts()` waits for the copy only when the sizes are needed, so a
    caller can launch the projection of view i+1 before it finishes view i (engine.RenderStep does)."""
    __slots__ = ("args", "bufs", "host_counts", "event", "D", "vis")


_pinned_pool = []


# "depth_major" (gs_isect_bin) or "emit_sort" (gs_isect_emit + gs_isect_sort: the upstream call shape, kept for callers that hold
# emitted keys; tests compare the two bit for bit)
BINNING = "depth_major"

def _project_stage(means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Optional[Tensor],
                   viewmat: Tensor, K: Tensor, W: int, H: int, tile_size: int, eps2d: float, near: float,
                   far: float, radius_clip: float) -> _Projected:
    lib = L.lib()
    dev = means.device
    N, D = means.shape[0], (0 if colors is None else colors.shape[1])
    i32, f32 = torch.int32, torch.float32
    gids = torch.empty(N, dtype=i32, device=dev); radii = torch.empty(N, dtype=i32, device=dev)
    means2d = torch.empty(N, 2, dtype=f32, device=dev); depths = torch.empty(N, dtype=f32, device=dev)
    conics = torch.empty(N, 3, dtype=f32, device=dev); comps = torch.empty(N, dtype=f32, device=dev)
    opac_p = torch.empty(N, dtype=f32, device=dev)
    colors_p = torch.empty(N, D, dtype=f32, device=dev) if D > 0 else None
    tpg = torch.empty(N, dtype=i32, device=dev); cum = torch.empty(N, dtype=torch.int64, device=dev)
    ws_bytes = lib.gs_project_ws_bytes(N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    # the compositor's 64-byte per-visible records come straight out of the projection (colours ride along for D <= 3)
    vis = torch.empty(N, 16, dtype=f32, device=dev) if (0 < D <= 3 and N > 0) else None
    st = L.stream()
    L.check(lib.gs_project_fwd_vis(N, L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(opacities), L.ptr(colors), D,
                                   L.ptr(viewmat), L.ptr(K), W, H, tile_size, L.f32(eps2d), L.f32(near), L.f32(far),
                                   L.f32(radius_clip), L.ptr(gids), L.ptr(radii), L.ptr(means2d), L.ptr(depths),
                                   L.ptr(conics), L.ptr(comps), L.ptr(opac_p), L.ptr(colors_p), L.ptr(tpg), L.ptr(cum),
                                   None, L.ptr(vis), L.ptr(ws), C.c_size_t(ws_bytes), L.ptr(counts), st), "gs_project_fwd_vis")
    pr = _Projected()
    pr.host_counts = _pinned_pool.pop() if _pinned_pool else torch.empty(2, dtype=torch.int64).pin_memory()
    pr.host_counts.copy_(counts, non_blocking=True)          # 16 bytes, asynchronous
    pr.event = torch.cuda.Event()
    pr.event.record()
    pr.bufs = (gids, radii, means2d, depths, conics, comps, opac_p, colors_p, tpg, cum, counts)
    pr.vis = vis
    pr.args = (W, H, tile_size)
    pr.D = D
    return pr


def _bin_stage(pr: _Projected, depth_channel: bool = False):
    """A2 emit, A3 sort, A4 offsets for a projected view; waits for its (V, I) -- the one host sync of the forward
    (as upstream).  Returns the state dict without the compositor outputs."""
    lib = L.lib()
    W, H, tile_size = pr.args
    gids, radii, means2d, depths, conics, comps, opac_p, colors_p, tpg, cum, _ = pr.bufs
    dev = means2d.device
    D = pr.D
    tw, th = (W + tile_size - 1) // tile_size, (H + tile_size - 1) // tile_size
    i32 = torch.int32
    st = L.stream()
    pr.event.synchronize()
    V, I = (int(x) for x in pr.host_counts.tolist())
    _pinned_pool.append(pr.host_counts)
    if V < 0 or I < 0 or I >= 2 ** 31:
        raise L.GeoSplatHipError(f"bad intersection count V={V} I={I}")
    gids, radii, means2d, depths = gids[:V], radii[:V], means2d[:V], depths[:V]
    conics, comps, opac_p, tpg, cum = conics[:V], comps[:V], opac_p[:V], tpg[:V], cum[:V]
    colors_p = colors_p[:V] if colors_p is not None else None
    if depth_channel:
        colors_p = depths.unsqueeze(-1).clone() if colors_p is None else torch.cat((colors_p, depths.unsqueeze(-1)), dim=1)
        D = D + 1
    colors_p = colors_p.contiguous()

    ids_s = torch.empty(I, dtype=torch.int64, device=dev); flat_s = torch.empty(I, dtype=i32, device=dev)
    if BINNING == "emit_sort":                                                   # the upstream call shape: emit, then sort the pairs
        ids = torch.empty(I, dtype=torch.int64, device=dev); flat = torch.empty(I, dtype=i32, device=dev)
        L.check(lib.gs_isect_emit(V, L.ptr(means2d), L.ptr(radii), L.ptr(depths), L.ptr(cum), tile_size, tw, th,
                                  L.ptr(ids), L.ptr(flat), st), "gs_isect_emit")
        sort_bytes = lib.gs_sort_ws_bytes(L.i64(I), tw, th)
        sort_ws = torch.empty(max(sort_bytes, 1), dtype=torch.uint8, device=dev)
        L.check(lib.gs_isect_sort(L.i64(I), L.ptr(ids), L.ptr(flat), L.ptr(ids_s), L.ptr(flat_s), tw, th, L.ptr(sort_ws),
                                  C.c_size_t(sort_bytes), st), "gs_isect_sort")
    else:                                                                        # depth-major binning (csrc/gs_sort.hip)
        bin_bytes = lib.gs_isect_bin_ws_bytes(V, L.i64(I), tw, th)
        bin_ws = torch.empty(max(bin_bytes, 1), dtype=torch.uint8, device=dev)
        L.check(lib.gs_isect_bin(V, L.ptr(means2d), L.ptr(radii), L.ptr(depths), L.ptr(tpg), L.i64(I), tile_size, tw, th,
                                 L.ptr(ids_s), L.ptr(flat_s), L.ptr(bin_ws), C.c_size_t(bin_bytes), st), "gs_isect_bin")
    offsets = torch.empty(th * tw, dtype=i32, device=dev)
    L.check(lib.gs_isect_offsets(L.i64(I), L.ptr(ids_s), tw * th, L.ptr(offsets), st), "gs_isect_offsets")
    state = dict(gaussian_ids_i32=gids, radii=radii, means2d=means2d, depths=depths, conics=conics,
                 compensations=comps, opacities=opac_p, colors=colors_p, tiles_per_gauss=tpg, isect_ids=ids_s,
                 flatten_ids=flat_s, isect_offsets=offsets)
    if pr.vis is not None and not depth_channel:             # (a depth channel changes the composited colours after the projection)
        state["vis_records"] = pr.vis
    return state, V, I, D, (W, H, tile_size)


def _prepare_stage(state, V: int, I: int, D: int, whs):
    """A5 preparation (per-visible records, sorted record stream, tile order) on the current stream."""
    lib = L.lib()
    W, H, tile_size = whs
    rws_bytes = lib.gs_raster_ws_bytes(L.i64(I), V, W, H, tile_size)
    rws = torch.empty(rws_bytes, dtype=torch.uint8, device=state["means2d"].device)   # written here, read by fwd AND bwd
    vis = state.get("vis_records")
    if vis is not None:
        L.check(lib.gs_raster_prepare_vis(W, H, tile_size, D, V, L.ptr(vis), L.i64(I), L.ptr(state["isect_offsets"]),
                                          L.ptr(state["flatten_ids"]), L.ptr(rws), C.c_size_t(rws_bytes), L.stream()),
                "gs_raster_prepare_vis")
    else:
        L.check(lib.gs_raster_prepare(W, H, tile_size, D, V, L.ptr(state["means2d"]), L.ptr(state["conics"]),
                                      L.ptr(state["opacities"]), L.ptr(state["colors"]), L.i64(I),
                                      L.ptr(state["isect_offsets"]), L.ptr(state["flatten_ids"]), L.ptr(rws),
                                      C.c_size_t(rws_bytes), L.stream()), "gs_raster_prepare")
    return dict({k: v for k, v in state.items() if k != "vis_records"}, raster_ws=rws)


def _composite_stage(state, V: int, I: int, D: int, whs, background: Optional[Tensor]):
    """A5 proper (the compositor) on the current stream; prepares the workspace first if the caller has not."""
    lib = L.lib()
    W, H, tile_size = whs
    dev = state["means2d"].device
    i32, f32 = torch.int32, torch.float32
    if "raster_ws" not in state:
        state = _prepare_stage(state, V, I, D, whs)
    rws = state["raster_ws"]
    render = torch.empty(H, W, D, dtype=f32, device=dev); alphas = torch.empty(H, W, dtype=f32, device=dev)
    last_ids = torch.empty(H, W, dtype=i32, device=dev)
    L.check(lib.gs_raster_composite(W, H, tile_size, D, V, L.ptr(state["colors"]), L.ptr(background), L.i64(I),
                                    L.ptr(state["isect_offsets"]), L.ptr(render), L.ptr(alphas), L.ptr(last_ids),
                                    L.ptr(rws), C.c_size_t(rws.numel()), L.stream()), "gs_raster_composite")
    state = dict(state, last_ids=last_ids)
    return render, alphas, state, V, I


# ---- capacity protocol (include/geosplat_hip.h): the same three stages without the (V, I) read-back --------------------------
def _bin_stage_cap(pr: _Projected, I_cap: int, status: Tensor, want_ids: bool = True):
    """A2-A4 with the counts left on the device: buffers, workspaces and grids are sized by (N, I_cap), the kernels read
    (V, I) from pr's device counts; an overflow is reported in `status` (int64[3], sticky), never written out of bounds."""
    lib = L.lib()
    W, H, tile_size = pr.args
    gids, radii, means2d, depths, conics, comps, opac_p, colors_p, tpg, cum, counts = pr.bufs
    dev = means2d.device
    N = means2d.shape[0]
    tw, th = (W + tile_size - 1) // tile_size, (H + tile_size - 1) // tile_size
    st = L.stream()
    flat_s = torch.empty(I_cap, dtype=torch.int32, device=dev)
    bin_bytes = lib.gs_isect_bin_ws_bytes(N, L.i64(I_cap), tw, th)
    bin_ws = torch.empty(max(bin_bytes, 1), dtype=torch.uint8, device=dev)
    offsets = torch.empty(th * tw, dtype=torch.int32, device=dev)
    if want_ids:
        ids_s = torch.empty(I_cap, dtype=torch.int64, device=dev)
        L.check(lib.gs_isect_bin_cap(N, L.ptr(means2d), L.ptr(radii), L.ptr(depths), L.ptr(counts), L.i64(I_cap), tile_size, tw, th,
                                     L.ptr(ids_s), L.ptr(flat_s), L.ptr(bin_ws), C.c_size_t(bin_bytes), L.ptr(status), st),
                "gs_isect_bin_cap")
        L.check(lib.gs_isect_offsets_cap(L.i64(I_cap), L.ptr(counts), L.ptr(ids_s), tw * th, L.ptr(offsets), st), "gs_isect_offsets_cap")
    else:                      # the caller only composites: int32 tile ids instead of the 64-bit keys (the `isect_ids` entry holds them)
        ids_s = torch.empty(I_cap, dtype=torch.int32, device=dev)
        L.check(lib.gs_isect_bin_tiles_cap(N, L.ptr(means2d), L.ptr(radii), L.ptr(depths), L.ptr(counts), L.i64(I_cap), tile_size, tw, th,
                                           L.ptr(ids_s), L.ptr(flat_s), L.ptr(bin_ws), C.c_size_t(bin_bytes), L.ptr(status), st),
                "gs_isect_bin_tiles_cap")
        L.check(lib.gs_isect_offsets_tiles_cap(L.i64(I_cap), L.ptr(counts), L.ptr(ids_s), tw * th, L.ptr(offsets), st),
                "gs_isect_offsets_tiles_cap")
    state = dict(gaussian_ids_i32=gids, radii=radii, means2d=means2d, depths=depths, conics=conics, compensations=comps,
                 opacities=opac_p, colors=colors_p, tiles_per_gauss=tpg, isect_ids=ids_s, flatten_ids=flat_s,
                 isect_offsets=offsets, counts=counts, vis_records=pr.vis)
    return state, N, I_cap, pr.D, (W, H, tile_size)


def _prepare_stage_cap(state, V_cap: int, I_cap: int, D: int, whs):
    lib = L.lib()
    W, H, tile_size = whs
    rws_bytes = lib.gs_raster_ws_bytes(L.i64(I_cap), V_cap, W, H, tile_size)
    rws = torch.empty(rws_bytes, dtype=torch.uint8, device=state["means2d"].device)
    L.check(lib.gs_raster_prepare_vis_cap(W, H, tile_size, D, V_cap, L.ptr(state["vis_records"]), L.i64(I_cap), L.ptr(state["counts"]),
                                          L.ptr(state["isect_offsets"]), L.ptr(state["flatten_ids"]), L.ptr(rws),
                                          C.c_size_t(rws_bytes), L.stream()), "gs_raster_prepare_vis_cap")
    return dict({k: v for k, v in state.items() if k != "vis_records"}, raster_ws=rws)


def _composite_stage_cap(state, V_cap: int, I_cap: int, D: int, whs, background: Optional[Tensor]):
    lib = L.lib()
    W, H, tile_size = whs
    dev = state["means2d"].device
    rws = state["raster_ws"]
    render = torch.empty(H, W, D, dtype=torch.float32, device=dev); alphas = torch.empty(H, W, dtype=torch.float32, device=dev)
    last_ids = torch.empty(H, W, dtype=torch.int32, device=dev)
    L.check(lib.gs_raster_composite_cap(W, H, tile_size, D, V_cap, L.ptr(state["colors"]), L.ptr(background), L.i64(I_cap),
                                        L.ptr(state["counts"]), L.ptr(state["isect_offsets"]), L.ptr(render), L.ptr(alphas),
                                        L.ptr(last_ids), L.ptr(rws), C.c_size_t(rws.numel()), L.stream()), "gs_raster_composite_cap")
    return render, alphas, dict(state, last_ids=last_ids)


def _raster_stage(pr: _Projected, background: Optional[Tensor], depth_channel: bool = False):
    state, V, I, D, whs = _bin_stage(pr, depth_channel)
    return _composite_stage(state, V, I, D, whs, background)


def _forward_stages(means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Optional[Tensor],
                    viewmat: Tensor, K: Tensor, W: int, H: int, tile_size: int, eps2d: float, near: float,
                    far: float, radius_clip: float, background: Optional[Tensor], depth_channel: bool = False):
    """A1..A5 through the C-ABI.  depth_channel appends the camera-space depth of every visible Gaussian as one
    more composited channel (gsplat render modes 'D' / 'ED' / 'RGB+D' / 'RGB+ED'; colors may be None for 'D'/'ED')."""
    pr = _project_stage(means, quats, scales, opacities, colors, viewmat, K, W, H, tile_size, eps2d, near, far, radius_clip)
    return _raster_stage(pr, background, depth_channel)


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, opacities, colors, viewmat, K, background, W, H, tile_size, eps2d, near, far,
                radius_clip, depth_channel):
        means, quats, scales = means.contiguous(), quats.contiguous(), scales.contiguous()
        opacities = opacities.contiguous()
        colors = colors.contiguous() if colors is not None else None
        viewmat, K = viewmat.contiguous(), K.contiguous()
        render, alphas, st, V, I = _forward_stages(means, quats, scales, opacities, colors, viewmat, K, W, H, tile_size,
                                                   eps2d, near, far, radius_clip, background, depth_channel)
        ctx.cfg = (W, H, tile_size, eps2d, V, I, depth_channel)
        ctx.save_for_backward(means, quats, scales, opacities, colors, viewmat, K, background, alphas,
                              *[st[k] for k in _META_KEYS])
        outs = (render, alphas) + tuple(st[k] for k in _META_KEYS)
        ctx.mark_non_differentiable(*outs[2:])
        return outs

    @staticmethod
    def backward(ctx, v_render, v_alphas, *_unused):
        lib = L.lib()
        W, H, tile_size, eps2d, V, I, depth_channel = ctx.cfg
        (means, quats, scales, opacities, colors, viewmat, K, background, alphas, *meta) = ctx.saved_tensors
        st = dict(zip(_META_KEYS, meta))
        dev = means.device
        N = means.shape[0]
        Dc = 0 if colors is None else colors.shape[1]       # input colour channels
        D = Dc + (1 if depth_channel else 0)                # composited channels
        f32 = torch.float32
        v_render = torch.zeros(H, W, D, dtype=f32, device=dev) if v_render is None else v_render.contiguous()
        v_alphas = torch.zeros(H, W, dtype=f32, device=dev) if v_alphas is None else v_alphas.contiguous()
        stride = lib.gs_raster_grad_stride(D)
        v_packed = torch.empty(V, stride, dtype=f32, device=dev)      # {v_means2d, v_conics, v_opacity, v_colors} per Gaussian
        s = L.stream()
        rws = st["raster_ws"]
        L.check(lib.gs_raster_bwd(W, H, tile_size, D, V, L.ptr(st["colors"]), L.ptr(background), L.i64(I),
                                  L.ptr(st["isect_offsets"]), L.ptr(alphas), L.ptr(st["last_ids"]), L.ptr(v_render),
                                  L.ptr(v_alphas), L.ptr(v_packed), L.ptr(rws), C.c_size_t(rws.numel()), s), "gs_raster_bwd")
        g_means = torch.empty(N, 3, dtype=f32, device=dev); g_quats = torch.empty(N, 4, dtype=f32, device=dev)
        g_scales = torch.empty(N, 3, dtype=f32, device=dev); g_opac = torch.empty(N, dtype=f32, device=dev)
        g_colors = torch.empty(N, Dc, dtype=f32, device=dev) if Dc > 0 else None
        v_depths = v_packed[:, 6 + Dc].contiguous() if depth_channel else None   # grad of the appended depth channel
        L.check(lib.gs_project_bwd(N, V, Dc, L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(opacities), L.ptr(viewmat),
                                   L.ptr(K), W, H, L.f32(eps2d), L.ptr(st["gaussian_ids_i32"]), L.ptr(st["conics"]),
                                   L.ptr(st["compensations"]), L.ptr(v_packed), stride, L.ptr(v_depths), L.ptr(g_means),
                                   L.ptr(g_quats), L.ptr(g_scales), L.ptr(g_opac), L.ptr(g_colors), 0, s), "gs_project_bwd")
        return (g_means, g_quats, g_scales, g_opac, g_colors) + (None,) * 11


def rasterization(means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor, viewmats: Tensor,
                  Ks: Tensor, width: int, height: int, near_plane: float = 0.01, far_plane: float = 1e10,
                  radius_clip: float = 0.0, eps2d: float = 0.3, sh_degree: Optional[int] = None, packed: bool = True,
                  tile_size: int = 16, backgrounds: Optional[Tensor] = None, render_mode: str = "RGB",
                  sparse_grad: bool = False, absgrad: bool = False, rasterize_mode: str = "antialiased",
                  ) -> Tuple[Tensor, Tensor, Dict]:
    """``gsplat.rasterization`` for the configuration the reference uses (one camera, packed, 'antialiased',
    'RGB', no SH).  Unsupported options raise (same error behaviour as upstream: Python exceptions)."""
    L.require_cuda(means, quats, scales, opacities, colors, viewmats, Ks)
    if viewmats.shape[:-2] != (1,) or Ks.shape[:-2] != (1,):
        raise ValueError("exactly one camera per call (rfstudio/model/gsplat.py:293 asserts cameras.shape == (1,))")
    if sh_degree is not None:
        raise NotImplementedError("sh_degree must be None (the reference passes sh_degree=None, geosplat uses sh_degree=0)")
    if render_mode not in ("RGB", "D", "ED", "RGB+D", "RGB+ED"):
        raise ValueError(f"unknown render_mode {render_mode!r}")
    if rasterize_mode != "antialiased":
        raise NotImplementedError("rasterize_mode 'antialiased' only (rfstudio/model/geosplat.py:796)")
    if sparse_grad or absgrad:
        raise NotImplementedError("sparse_grad / absgrad are False on the reference path")
    N = means.shape[0]
    assert means.shape == (N, 3) and quats.shape == (N, 4) and scales.shape == (N, 3) and opacities.shape == (N,)
    depth_channel = render_mode != "RGB"
    if render_mode in ("D", "ED"):
        colors = None                                    # upstream ignores colours in the depth-only modes
    else:
        assert colors.dim() == 2 and colors.shape[0] == N
    bg = None
    if backgrounds is not None:
        bg = backgrounds.reshape(-1).contiguous().float()
        if depth_channel:                                # upstream pads the background of the depth channel with 0
            bg = torch.cat((bg, bg.new_zeros(1))) if colors is not None else bg.new_zeros(1)
        assert bg.shape[0] == (0 if colors is None else colors.shape[1]) + (1 if depth_channel else 0)
    outs = _Rasterize.apply(means.float(), quats.float(), scales.float(), opacities.float(),
                            None if colors is None else colors.float(),
                            viewmats.reshape(4, 4).float(), Ks.reshape(3, 3).float(), bg, int(width), int(height),
                            int(tile_size), float(eps2d), float(near_plane), float(far_plane), float(radius_clip),
                            depth_channel)
    render, alphas = outs[0], outs[1]
    if render_mode in ("ED", "RGB+ED"):                  # expected depth = accumulated depth / alpha (rfstudio/model/gsplat.py:151-172)
        render = torch.cat((render[..., :-1], render[..., -1:] / alphas.unsqueeze(-1).clamp(min=1e-10)), dim=-1)
    st = dict(zip(_META_KEYS, outs[2:]))
    tw, th = (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size
    V = st["radii"].shape[0]
    meta = {
        "camera_ids": torch.zeros(V, dtype=torch.int64, device=means.device),
        "gaussian_ids": st["gaussian_ids_i32"].long(),
        "radii": st["radii"], "means2d": st["means2d"], "depths": st["depths"], "conics": st["conics"],
        "opacities": st["opacities"], "tile_width": tw, "tile_height": th, "tiles_per_gauss": st["tiles_per_gauss"],
        "isect_ids": st["isect_ids"], "flatten_ids": st["flatten_ids"],
        "isect_offsets": st["isect_offsets"].view(1, th, tw), "width": width, "height": height,
        "tile_size": tile_size, "n_cameras": 1,
        # extras (not in upstream's dict)
        "compensations": st["compensations"], "last_ids": st["last_ids"].view(1, height, width),
    }
    return render.unsqueeze(0), alphas.view(1, height, width, 1), meta
