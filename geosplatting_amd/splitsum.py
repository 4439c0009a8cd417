"""Split-sum environment pyramid (stage S5) -- host-side mirror of
``TextureCubeMap.as_splitsum`` / ``TextureSplitSum`` (rfstudio/graphics/_mesh/_texture.py:530-613) and of the
plugin wrappers in rfstudio/graphics/_mesh/_splitsum/_wrap.py:82-157.  All arithmetic runs in
libgeosplat_hip.so; this file only owns tensors, the bounds cache and the autograd glue.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib as L


# ----------------------------------------------------------------------------- cutoff + bounds cache
_cutoff_cache: Dict[Tuple[float, float], float] = {}
_bounds_cache: Dict[Tuple[int, float, float, int], Tuple[float, Tensor]] = {}


def ndf_cutoff(roughness: float, cutoff: float = 0.99, n_samples: int = 1000000) -> float:
    """cos(theta) retaining `cutoff` of the GGX NDF energy -- float64 numpy CDF exactly like the reference
    (rfstudio/graphics/_mesh/_splitsum/_wrap.py:120-135)."""
    key = (float(roughness), float(cutoff))
    if key not in _cutoff_cache:
        costheta = np.cos(np.linspace(0, np.pi / 2.0, n_samples))
        c = np.clip(costheta, 0.0, 1.0)
        a2 = roughness ** 4
        d = (c * a2 - c) * c + 1.0
        D = np.cumsum(a2 / (d * d * np.pi))
        _cutoff_cache[key] = float(costheta[np.argmax(D >= D[-1] * cutoff)])
    return _cutoff_cache[key]


def specular_bounds(res: int, roughness: float, cutoff: float, device: torch.device) -> Tuple[float, Tensor]:
    """Per-texel AABBs of the lobe footprint, cached per (res, roughness, cutoff, device) like
    _wrap.py:136,151-154."""
    key = (res, float(roughness), float(cutoff), device.index or 0)
    if key not in _bounds_cache:
        ct = ndf_cutoff(roughness, cutoff)
        b = torch.empty(6, res, res, 24, dtype=torch.float32, device=device)
        # (gs_specular_bounds is the kernel shaped like the reference's SpecularBoundsKernel, 0.2 s at 512^2; tests compare the two)
        nbytes = L.lib().gs_specular_bounds_ws_bytes(res)
        ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
        L.check(L.lib().gs_specular_bounds_fast(res, L.f32(ct), L.ptr(dir_table(res, device)), L.ptr(b), L.ptr(ws), C.c_size_t(nbytes),
                                                L.stream()), "gs_specular_bounds_fast")
        _bounds_cache[key] = (ct, b)
    return _bounds_cache[key]


_dir_table_cache: Dict[Tuple[int, int], Tensor] = {}


def dir_table(res: int, device: torch.device) -> Tensor:
    """[6,R,R,4] = (unit direction, pixel_area) of every texel; a function of R only, cached like the bounds."""
    key = (res, device.index or 0)
    if key not in _dir_table_cache:
        t = torch.empty(6, res, res, 4, dtype=torch.float32, device=device)
        L.check(L.lib().gs_cube_dir_table(res, L.ptr(t), L.stream()), "gs_cube_dir_table")
        _dir_table_cache[key] = t
    return _dir_table_cache[key]


# ----------------------------------------------------------------------------- tiled pair-weight tables (levels with R >= 64)
TILED_PREFILTER = True            # False (tests): every level through the direct kernels
BWD_MARGIN = 2                    # candidate outputs of a source texel: its own lobe box grown by this (csrc GS_SPECULAR_BWD_MARGIN)
_ROW_PAD = 8                      # csrc GS_TILE_ROW_PAD


def _bwd_margin(res: int) -> int:
    """levels below 64^2 have one or four 16x16 culling tiles per face -- the reference's boxes are far from symmetric there:
    every texel of every face is a candidate output (margin >= R), the pair test decides"""
    return BWD_MARGIN if res >= 64 else res


def tile_geometry(res: int) -> Tuple[int, int]:
    """(bw, nb): blocks per tile row, blocks per tile, for the 16 waves of a workgroup.  Large tiles where the lobes are small
    against the face (the staged rectangle grows by the lobe diameter once per tile), one 8x8 block split over all 16 waves
    where a lobe covers most of a face and the level has few texels."""
    # measured per level with scripts/prefilter_bench.py (profiles/r03_prefilter_geometry.txt)
    if res >= 512:
        return 4, 16          # 32 x 32 texels, one wave per block (32x16: 217 us instead of 161; 16x16: 273)
    if res >= 128:
        return 2, 4           # 16 x 16, four waves per block (256^2: 32x16 307 us, 32x32 369; 128^2: 32x16 180, 32x32 331)
    if res >= 64:
        return 1, 2           # 8 x 16, eight waves per block (8x8: 54 us instead of 47; 16x16: 76)
    return 1, 1               # 8 x 8, sixteen waves on the block
_tiles_cache: Dict[Tuple[int, float, float, int], Optional[Dict]] = {}


def tiles_eligible(res: int) -> bool:
    return TILED_PREFILTER and res >= 16 and res % 16 == 0


def _build_direction(res: int, roughness: float, ct: float, bounds: Tensor, table: Tensor, tiles: Tensor, backward: int, bw: int,
                     nb: int) -> Dict:
    lib = L.lib()
    dev = bounds.device
    nt = tiles.shape[0]
    ns = 6 * nb                                                            # slots per tile
    margin = _bwd_margin(res)
    cnt = torch.zeros(nt * ns, dtype=torch.int32, device=dev)
    ext = torch.zeros(nt * ns, 4, dtype=torch.int32, device=dev)
    pairs = torch.zeros(1, dtype=torch.int64, device=dev)
    L.check(lib.gs_specular_tiles_count(res, L.f32(roughness), L.f32(ct), backward, margin, bw, nb, L.ptr(bounds), L.ptr(table),
                                        L.ptr(tiles), nt, L.ptr(cnt), L.ptr(ext), L.ptr(pairs), L.stream()), "gs_specular_tiles_count")
    cnt_pad = ((cnt.long() + (_ROW_PAD - 1)) // _ROW_PAD) * _ROW_PAD
    csum = torch.cumsum(cnt_pad, 0)
    row_begin = (csum - cnt_pad).contiguous()
    total = int(csum[-1].item())
    # per (tile, face): the source rectangle its four blocks can address, staged in LDS with a pitch = 8 (mod 16) texels
    # (conflict-free ds_read_b128 for the 8x8 lane arrangement: 16-lane groups cover 64 distinct banks)
    c4 = cnt.view(nt, 6, nb)
    e4 = ext.view(nt, 6, nb, 4).long()
    on = c4 > 0
    big = 1 << 30
    x0 = torch.where(on, e4[..., 0], torch.full_like(e4[..., 0], big)).amin(-1)
    x1 = torch.where(on, e4[..., 1], torch.full_like(e4[..., 1], -big)).amax(-1)
    y0 = torch.where(on, e4[..., 2], torch.full_like(e4[..., 2], big)).amin(-1)
    y1 = torch.where(on, e4[..., 3], torch.full_like(e4[..., 3], -big)).amax(-1)
    some = on.any(-1)
    rw = torch.where(some, x1 - x0 + 1, torch.zeros_like(x0))
    rh = torch.where(some, y1 - y0 + 1, torch.zeros_like(x0))
    pitch = torch.where(some, rw + ((8 - rw) % 16), torch.zeros_like(rw))
    seg = torch.stack((torch.where(some, x0, torch.zeros_like(x0)), torch.where(some, y0, torch.zeros_like(y0)), rh, pitch), -1).int().contiguous()
    lds_bytes = int((rh * pitch).max().item()) * 16
    desc = torch.zeros(max(total, 2 * _ROW_PAD), dtype=torch.int32, device=dev)
    weights = torch.zeros(max(total, 2 * _ROW_PAD) * 64, dtype=torch.float32, device=dev)
    L.check(lib.gs_specular_tiles_fill(res, L.f32(roughness), L.f32(ct), backward, margin, bw, nb, L.ptr(bounds), L.ptr(table),
                                       L.ptr(tiles), nt, L.ptr(row_begin), L.ptr(seg), L.ptr(desc), L.ptr(weights), L.stream()),
            "gs_specular_tiles_fill")
    rows_per_tile = cnt_pad.view(nt, ns).sum(-1)
    return {"cnt": cnt_pad.int().view(nt, ns).contiguous(), "row_begin": row_begin.view(nt, ns).contiguous(), "seg": seg, "desc": desc,
            "weights": weights, "rows": total, "kept_rows": int(cnt.sum().item()), "pairs": int(pairs.item()), "lds_bytes": lds_bytes,
            "rows_per_tile": rows_per_tile}


def specular_tiles(res: int, roughness: float, cutoff: float, device: torch.device) -> Optional[Dict]:
    """Tiled pair-weight tables of one pyramid level (csrc/gs_splitsum_tiles.hip), built once per (res, roughness, cutoff, device)
    -- they do not depend on the cubemap.  With exact mirror symmetry (checked on the device) the tables cover ONE octant of
    the cube and every row serves eight reflections: ~1.4 GB for a 512^2 pyramid (round 2's per-texel tables: 13 GB).  Returns
    None when the level is not eligible (R not a multiple of 16: the direct kernels run)."""
    key = (res, float(roughness), float(cutoff), device.index or 0)
    if key in _tiles_cache:
        return _tiles_cache[key]
    if not tiles_eligible(res):
        _tiles_cache[key] = None
        return None
    lib = L.lib()
    ct, bounds = specular_bounds(res, roughness, cutoff, device)
    table = dir_table(res, device)
    chk = torch.zeros(2, dtype=torch.int64, device=device)
    L.check(lib.gs_specular_tiles_check(res, L.ptr(bounds), L.ptr(table), L.ptr(chk), L.stream()), "gs_specular_tiles_check")
    bad_dir, bad_box = (int(v) for v in chk.tolist())
    bw, nb = tile_geometry(res)
    tw, th = 8 * bw, 8 * (nb // bw)                                       # tile size in texels
    half = res // 2
    sym = (bad_dir == 0 and bad_box == 0 and half % tw == 0 and half % th == 0 and os.environ.get("GEOSPLAT_PREFILTER_SYMMETRY", "1") != "0")
    if sym:      # one quadrant of the faces +x, +y, +z = one texel of every orbit of the three reflections
        coords = [(s, tw * i, th * j) for s in (0, 2, 4) for j in range(half // th) for i in range(half // tw)]
    else:
        if res % tw or res % th:
            bw, nb = 1, 1
            tw = th = 8
        coords = [(s, tw * i, th * j) for s in range(6) for j in range(res // th) for i in range(res // tw)]
    tiles = torch.tensor([(s, x, y, 0) for s, x, y in coords], dtype=torch.int32, device=device)
    entry = {"res": res, "ct": ct, "bounds": bounds, "table": table, "tiles": tiles, "n_tiles": len(coords), "n_mirrors": 8 if sym else 1,
             "symmetry_check": (bad_dir, bad_box), "orders": {}, "bw": bw, "nb": nb, "tile_texels": tw * th}
    for name, bwd in (("fwd", 0), ("bwd", 1)):
        entry[name] = _build_direction(res, roughness, ct, bounds, table, tiles, bwd, bw, nb)
    # every pair of the forward tables must appear exactly once in the transposed ones
    if entry["fwd"]["pairs"] != entry["bwd"]["pairs"]:
        raise L.GeoSplatHipError(f"specular tiles R={res}: {entry['fwd']['pairs']} forward pairs but {entry['bwd']['pairs']} transposed "
                                 f"pairs (candidate margin {BWD_MARGIN} too small?)")
    # sum of the pair weights per output texel (needs pixel_area of the actual sources: not mirror symmetric) -- the direct kernel on ones
    ones = torch.ones(6, res, res, 3, dtype=torch.float32, device=device)
    raw = torch.empty(6, res, res, 4, dtype=torch.float32, device=device)
    L.check(lib.gs_specular_cubemap_fwd(res, L.ptr(ones), L.ptr(bounds), L.ptr(table), L.f32(roughness), L.f32(ct), L.ptr(raw), L.stream()),
            "gs_specular_cubemap_fwd")
    entry["wsum"] = raw[..., 3].contiguous()
    entry["inv_wsum"] = (1.0 / entry["wsum"]).contiguous()                # scale of the staged d loss / d level texels (backward)
    entry["area4"] = (table[..., 3] * 0.25).contiguous()                  # pixel_area / 4: scale of the staged sources (forward)
    _tiles_cache[key] = entry
    return entry


def shard_tiles(n_tiles: int, rank: int, world: int) -> Tuple[int, int]:
    """Share [begin, end) of a level's tile list for `rank`: the list is the longest-first order dealt round-robin (rank r gets
    sorted tiles r, r + world, ...), stored rank after rank -- contiguous shares whose sizes differ by at most one."""
    sizes = [(n_tiles - q + world - 1) // world for q in range(world)]
    begin = sum(sizes[:rank])
    return begin, begin + sizes[rank]


def _ordered(entry: Dict, direction: str, world: int) -> Dict:
    """Tile metadata of one direction in launch order: longest tile first (the lobes of tiles at the cube's corners are 3x the
    mean); for `world` ranks the sorted list is dealt round-robin and stored share after share (shard_tiles)."""
    key = (direction, world)
    if key not in entry["orders"]:
        d = entry[direction]
        order = torch.argsort(d["rows_per_tile"], descending=True, stable=True)
        if world > 1:
            order = torch.cat([order[r::world] for r in range(world)])
        entry["orders"][key] = {"tiles": entry["tiles"][order].contiguous(), "seg": d["seg"][order].contiguous(),
                                "row_begin": d["row_begin"][order].contiguous(), "cnt": d["cnt"][order].contiguous()}
    return entry["orders"][key]


def _tiles_apply(entry: Dict, direction: str, src: Tensor, dst: Tensor, tile_begin: int = 0, tile_end: Optional[int] = None,
                 world: int = 1) -> None:
    d = entry[direction]
    o = _ordered(entry, direction, world)
    te = entry["n_tiles"] if tile_end is None else tile_end
    bwd = direction == "bwd"
    L.check(L.lib().gs_specular_tiles_apply(entry["res"], 1 if bwd else 0, entry["n_mirrors"], _bwd_margin(entry["res"]), entry["bw"],
                                            entry["nb"], L.ptr(src), L.ptr(entry["inv_wsum"] if bwd else entry["area4"]),
                                            L.ptr(entry["area4"]) if bwd else None, L.ptr(entry["bounds"]), L.ptr(o["tiles"]),
                                            L.ptr(o["seg"]), L.ptr(o["row_begin"]), L.ptr(o["cnt"]), L.ptr(d["desc"]), L.ptr(d["weights"]),
                                            L.ptr(dst), tile_begin, te, C.c_size_t(d["lds_bytes"]), L.stream()), "gs_specular_tiles_apply")


MERGED_APPLY = True               # False (tests): one launch per level, the launch sequence of rounds 3-5


def _tiles_apply_multi(jobs, direction: str, world: int = 1) -> None:
    """All levels of one direction in ONE launch (gs_specular_tiles_apply_multi): jobs = [(entry, src, dst, tile_begin, tile_end)].
    Falls back to one launch per level when the levels do not share n_mirrors (or MERGED_APPLY is off)."""
    jobs = [j for j in jobs if (j[4] if j[4] is not None else j[0]["n_tiles"]) > j[3]]
    if not jobs:
        return
    if not MERGED_APPLY or len(jobs) == 1 or len(jobs) > 8 or len({j[0]["n_mirrors"] for j in jobs}) != 1:
        for e, src, dst, t0, t1 in jobs:
            _tiles_apply(e, direction, src, dst, t0, t1, world)
        return
    # (measured and removed: two launches by LDS class -- levels that need less than half of a CU's LDS, two workgroups per CU, apart
    #  from the ones that need more: 0.710 + 0.747 ms against 0.696 + 0.728 for ONE launch at the largest level's LDS)
    bwd = direction == "bwd"
    arr = (L.GsTileLevel * len(jobs))()
    keep = []
    for k, (e, src, dst, t0, t1) in enumerate(jobs):
        d = e[direction]
        o = _ordered(e, direction, world)
        src = src.contiguous()
        keep.append(src)
        g = arr[k]
        g.R, g.n_mirrors, g.margin, g.bw, g.nb = e["res"], e["n_mirrors"], _bwd_margin(e["res"]), e["bw"], e["nb"]
        g.tile_begin, g.tile_end = t0, e["n_tiles"] if t1 is None else t1
        g.src = src.data_ptr(); g.scale = (e["inv_wsum"] if bwd else e["area4"]).data_ptr()
        g.out_scale = e["area4"].data_ptr() if bwd else None
        g.bounds = e["bounds"].data_ptr(); g.tiles = o["tiles"].data_ptr(); g.segments = o["seg"].data_ptr()
        g.row_begin = o["row_begin"].data_ptr(); g.row_counts = o["cnt"].data_ptr(); g.desc = d["desc"].data_ptr()
        g.weights = d["weights"].data_ptr(); g.dst = dst.data_ptr(); g.lds_bytes = d["lds_bytes"]
        assert dst.is_contiguous() and dst.is_cuda
    L.check(L.lib().gs_specular_tiles_apply_multi(len(jobs), arr, 1 if bwd else 0, L.stream()), "gs_specular_tiles_apply_multi")


def mip_chain(cubemap: Tensor, min_resolution: int = 16) -> List[Tensor]:
    """[cubemap, mip 1, ..., mip n] (n halvings down to min_resolution) without an autograd graph; one launch when n <= 5."""
    cubemap = cubemap.detach().float().contiguous()
    R = int(cubemap.shape[1])
    res = []
    r = R
    while r > min_resolution:
        r //= 2
        res.append(r)
    if not res:
        return [cubemap]
    if len(res) <= 5 and cubemap.shape[3] == 3 and R % (1 << len(res)) == 0:
        sizes = [6 * q * q * 3 for q in res]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=cubemap.device)
        outs = [t.view(6, q, q, 3) for t, q in zip(torch.split(flat, sizes), res)]
        ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        L.check(L.lib().gs_cubemap_mip_chain_fwd(R, len(outs), L.ptr(cubemap), ptrs, L.stream()), "gs_cubemap_mip_chain_fwd")
        return [cubemap] + outs
    mips = [cubemap]
    with torch.no_grad():
        while mips[-1].shape[1] > min_resolution:
            mips.append(_CubeMapMip.apply(mips[-1]))
    return mips


class _CubeMapMip(torch.autograd.Function):
    """rfstudio/graphics/_mesh/_texture.py:199-226"""

    @staticmethod
    def forward(ctx, cubemap: Tensor) -> Tensor:
        cubemap = cubemap.contiguous()
        R, Cn = cubemap.shape[1], cubemap.shape[3]
        out = torch.empty(6, R // 2, R // 2, Cn, dtype=torch.float32, device=cubemap.device)
        L.check(L.lib().gs_cubemap_mip_fwd(R, Cn, L.ptr(cubemap), L.ptr(out), L.stream()), "gs_cubemap_mip_fwd")
        return out

    @staticmethod
    def backward(ctx, dout: Tensor) -> Tensor:
        dout = dout.contiguous()
        R = dout.shape[1]
        assert dout.shape[3] == 3
        g = torch.empty(6, 2 * R, 2 * R, 3, dtype=torch.float32, device=dout.device)
        L.check(L.lib().gs_cubemap_mip_bwd(R, L.ptr(dout), L.ptr(g), 0, L.stream()), "gs_cubemap_mip_bwd")
        return g


class _DiffuseCubemap(torch.autograd.Function):
    """_diffuse_cubemap_func (rfstudio/graphics/_mesh/_splitsum/_wrap.py:82-93)"""

    @staticmethod
    def forward(ctx, cubemap: Tensor) -> Tensor:
        cubemap = cubemap.contiguous()
        out = torch.empty_like(cubemap)
        L.check(L.lib().gs_diffuse_cubemap_fwd(cubemap.shape[1], L.ptr(cubemap), L.ptr(out), L.stream()),
                "gs_diffuse_cubemap_fwd")
        return out

    @staticmethod
    def backward(ctx, dout: Tensor) -> Tensor:
        dout = dout.contiguous()
        g = torch.empty_like(dout)
        L.check(L.lib().gs_diffuse_cubemap_bwd(dout.shape[1], L.ptr(dout), L.ptr(g), 0, L.stream()),
                "gs_diffuse_cubemap_bwd")
        return g


class _SpecularCubemap(torch.autograd.Function):
    """_specular_cubemap + the rgb/wsum normalisation (rfstudio/graphics/_mesh/_splitsum/_wrap.py:104-118,157).
    Direct evaluation of the lobe weights (every call recomputes them)."""

    @staticmethod
    def forward(ctx, cubemap: Tensor, roughness: float, costheta_cutoff: float, bounds: Tensor) -> Tensor:
        cubemap = cubemap.contiguous()
        R = cubemap.shape[1]
        raw = torch.empty(6, R, R, 4, dtype=torch.float32, device=cubemap.device)
        table = dir_table(R, cubemap.device)
        L.check(L.lib().gs_specular_cubemap_fwd(R, L.ptr(cubemap), L.ptr(bounds), L.ptr(table), L.f32(roughness),
                                                L.f32(costheta_cutoff), L.ptr(raw), L.stream()),
                "gs_specular_cubemap_fwd")
        wsum = raw[..., 3:]
        ctx.save_for_backward(bounds, wsum)
        ctx.cfg = (roughness, costheta_cutoff)
        return raw[..., :3] / wsum

    @staticmethod
    def backward(ctx, dout: Tensor):
        bounds, wsum = ctx.saved_tensors
        roughness, ct = ctx.cfg
        v = (dout / wsum).contiguous()             # wsum does not depend on the cubemap
        g = torch.empty_like(v)
        table = dir_table(v.shape[1], v.device)
        L.check(L.lib().gs_specular_cubemap_bwd(v.shape[1], L.ptr(bounds), L.ptr(table), L.ptr(v), L.f32(roughness), L.f32(ct),
                                                L.ptr(g), 0, L.stream()), "gs_specular_cubemap_bwd")
        return g, None, None, None


class _SpecularCubemapTiled(torch.autograd.Function):
    """The same operator through the tiled pair-weight tables (specular_tiles): normalisation by the weight sums fused into the
    kernels, backward = the transposed tables as a gather."""

    @staticmethod
    def forward(ctx, cubemap: Tensor, res: int, roughness: float, cutoff: float) -> Tensor:
        e = specular_tiles(res, roughness, cutoff, cubemap.device)
        out = torch.empty(6, res, res, 3, dtype=torch.float32, device=cubemap.device)
        _tiles_apply(e, "fwd", cubemap.contiguous(), out)
        ctx.cfg = (res, roughness, cutoff)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        res, roughness, cutoff = ctx.cfg
        e = specular_tiles(res, roughness, cutoff, dout.device)
        g = torch.empty(6, res, res, 3, dtype=torch.float32, device=dout.device)
        _tiles_apply(e, "bwd", dout.contiguous(), g)
        return g, None, None, None


def diffuse_cubemap(cubemap: Tensor) -> Tensor:
    return _DiffuseCubemap.apply(cubemap)


def specular_cubemap(cubemap: Tensor, roughness: float, cutoff: float = 0.99, cached: Optional[bool] = None) -> Tensor:
    """specular_cubemap of _wrap.py:138-157.  cached=True (default where the level is eligible: R >= 64, a multiple of 16) applies
    the tiled pair-weight tables, cached=False evaluates every pair weight in the kernel (the reference's call shape)."""
    res = int(cubemap.shape[1])
    use_tables = tiles_eligible(res) if cached is None else (cached and tiles_eligible(res))
    if use_tables:
        return _SpecularCubemapTiled.apply(cubemap, res, float(roughness), float(cutoff))
    ct, bounds = specular_bounds(res, roughness, cutoff, cubemap.device)
    return _SpecularCubemap.apply(cubemap, float(roughness), ct, bounds)


def _specular_level_backward(gl: Tensor, rough: float, cutoff: float) -> Tensor:
    """d loss / d (mip level) from d loss / d (prefiltered level), on the current stream, without an autograd graph."""
    res = gl.shape[1]
    g = torch.empty(6, res, res, 3, dtype=torch.float32, device=gl.device)
    e = specular_tiles(res, rough, cutoff, gl.device)
    if e is not None:
        _tiles_apply(e, "bwd", gl.contiguous(), g)
        return g
    ct, bounds = specular_bounds(res, rough, cutoff, gl.device)
    wsum = _direct_wsum(res, rough, cutoff, gl.device)
    v = (gl / wsum).contiguous()
    L.check(L.lib().gs_specular_cubemap_bwd(res, L.ptr(bounds), L.ptr(dir_table(res, gl.device)), L.ptr(v), L.f32(rough), L.f32(ct),
                                            L.ptr(g), 0, L.stream()), "gs_specular_cubemap_bwd")
    return g


_wsum_cache: Dict[Tuple[int, float, float, int], Tensor] = {}


def _direct_wsum(res: int, roughness: float, cutoff: float, device: torch.device) -> Tensor:
    """[6,R,R,1] sums of the pair weights per output texel (they do not depend on the cubemap): the direct kernel on ones."""
    key = (res, float(roughness), float(cutoff), device.index or 0)
    if key not in _wsum_cache:
        ct, bounds = specular_bounds(res, roughness, cutoff, device)
        ones = torch.ones(6, res, res, 3, dtype=torch.float32, device=device)
        raw = torch.empty(6, res, res, 4, dtype=torch.float32, device=device)
        L.check(L.lib().gs_specular_cubemap_fwd(res, L.ptr(ones), L.ptr(bounds), L.ptr(dir_table(res, device)), L.f32(roughness),
                                                L.f32(ct), L.ptr(raw), L.stream()), "gs_specular_cubemap_fwd")
        _wsum_cache[key] = raw[..., 3:].contiguous()
    return _wsum_cache[key]


# ----------------------------------------------------------------------------- atlas packing (reference layout)
def merge_mipmaps(mipmaps: List[Tensor]) -> Tensor:
    """[6,R,R,3], [6,R/2,R/2,3], ... -> atlas [6,4,R,R] (rfstudio/graphics/_mesh/_texture.py:228-244)"""
    R = mipmaps[0].shape[-2]
    out = torch.stack((mipmaps[0][..., 0], mipmaps[0][..., 1], mipmaps[0][..., 2],
                       torch.zeros_like(mipmaps[0][..., 0])), dim=-3)
    origin = 0
    for i in range(1, len(mipmaps)):
        h = R // 2
        out[..., 3, origin:origin + h, origin:origin + h] = mipmaps[i][..., 0]
        out[..., 3, origin:origin + h, origin + h:origin + R] = mipmaps[i][..., 1]
        out[..., 3, origin + h:origin + R, origin:origin + h] = mipmaps[i][..., 2]
        origin += h
        R = h
    return out


def split_mipmaps(atlas: Tensor, num_mipmaps: int) -> List[Tensor]:
    """atlas [6,4,R,R] -> list of [6,R_l,R_l,3] (rfstudio/graphics/_mesh/_texture.py:247-261)"""
    res = []
    bs = atlas.shape[:-3]
    for _ in range(num_mipmaps):
        R = atlas.shape[-1]
        res.append(atlas[..., :3, :, :].flatten(-2, -1).transpose(-2, -1).reshape(*bs, R, R, 3).contiguous())
        h = R // 2
        atlas = atlas[..., 3, :, :].reshape(*bs, 2, h, 2, h).transpose(-3, -2).reshape(*bs, 4, h, h)
    return res


# ----------------------------------------------------------------------------- the pyramid object
@dataclass
class TextureSplitSum:
    """Field-compatible with rfstudio's TextureSplitSum (base, mipmaps atlas, num_mipmaps, min/max roughness);
    additionally keeps the per-level tensors so that the fused shading kernel reads them without an
    atlas round trip (the atlas is materialised lazily by ``.mipmaps``)."""
    base: Tensor                       # [6,16,16,3]
    levels: List[Tensor]               # L x [6,R_l,R_l,3]
    min_roughness: float = 0.08
    max_roughness: float = 0.5

    @property
    def num_mipmaps(self) -> int:
        return len(self.levels)

    @property
    def mipmaps(self) -> Tensor:
        return merge_mipmaps(self.levels)

    @classmethod
    def from_atlas(cls, base: Tensor, mipmaps: Tensor, num_mipmaps: int, min_roughness=0.08, max_roughness=0.5):
        return cls(base, split_mipmaps(mipmaps, num_mipmaps), float(min_roughness), float(max_roughness))


def _mip_chain_backward(g_mips: List[Tensor], g_base: Tensor) -> Tensor:
    """Diffuse backward and the mip links, each ADDING into the (fresh, contiguous) gradient of the level it feeds: the sums of the
    reference's autograd graph in the same order, without separate additions (accumulate flags of the two kernels)."""
    gd = g_base.contiguous()
    L.check(L.lib().gs_diffuse_cubemap_bwd(gd.shape[1], L.ptr(gd), L.ptr(g_mips[-1]), 1, L.stream()), "gs_diffuse_cubemap_bwd")
    for idx in range(len(g_mips) - 1, 0, -1):
        R = g_mips[idx].shape[1]
        L.check(L.lib().gs_cubemap_mip_bwd(R, L.ptr(g_mips[idx]), L.ptr(g_mips[idx - 1]), 1, L.stream()), "gs_cubemap_mip_bwd")
    return g_mips[0]


def as_splitsum_backward(g_base: Tensor, g_levels: List[Tensor], *, cutoff: float = 0.99, min_roughness: float = 0.08,
                         max_roughness: float = 0.5, out: Optional[Tensor] = None) -> Tensor:
    """Explicit backward of `as_splitsum` on the CURRENT stream (no autograd graph, so a caller can place it on any HIP
    stream -- the autograd engine would run it on the stream of the forward): texel gradients of the base map and of
    the n levels -> gradient of the cubemap (written into `out` [6,R,R,3] when given).  The tiled levels run as ONE launch."""
    n = len(g_levels)
    roughs = _level_roughness(n, min_roughness, max_roughness)
    g_mips: List[Optional[Tensor]] = [None] * n
    jobs = []
    for i, (gl, rough) in enumerate(zip(g_levels, roughs)):
        res = gl.shape[1]
        e = specular_tiles(res, rough, cutoff, gl.device)
        if e is None:
            g_mips[i] = _specular_level_backward(gl, rough, cutoff)
            continue
        g = out if (i == 0 and out is not None) else torch.empty(6, res, res, 3, dtype=torch.float32, device=gl.device)
        jobs.append((e, gl, g, 0, None))
        g_mips[i] = g
    _tiles_apply_multi(jobs, "bwd")
    if out is not None and g_mips[0] is not out:
        out.copy_(g_mips[0]); g_mips[0] = out
    return _mip_chain_backward(g_mips, g_base)


class _SplitSumFused(torch.autograd.Function):
    """as_splitsum as ONE autograd node: mip chain (one launch), diffuse map, every specular level (one launch); backward =
    as_splitsum_backward.  Used when every level goes through the tiled tables (R a multiple of 16 down to min_resolution)."""

    @staticmethod
    def forward(ctx, cubemap: Tensor, cutoff: float, min_resolution: int, min_roughness: float, max_roughness: float):
        base, levels = _as_splitsum_fused(cubemap, cutoff, min_resolution, min_roughness, max_roughness)
        ctx.cfg = (cutoff, min_roughness, max_roughness, [tuple(l.shape) for l in levels], tuple(base.shape))
        return (base, *levels)

    @staticmethod
    def backward(ctx, g_base, *g_levels):
        cutoff, min_roughness, max_roughness, shapes, bshape = ctx.cfg
        dev = next(g for g in (g_base, *g_levels) if g is not None).device
        g_base = torch.zeros(bshape, dtype=torch.float32, device=dev) if g_base is None else g_base.contiguous()
        gl = [torch.zeros(sh, dtype=torch.float32, device=dev) if g is None else g.contiguous() for g, sh in zip(g_levels, shapes)]
        g = as_splitsum_backward(g_base, gl, cutoff=cutoff, min_roughness=min_roughness, max_roughness=max_roughness)
        return g, None, None, None, None


def _fused_eligible(cubemap: Tensor, min_resolution: int) -> bool:
    R = int(cubemap.shape[1])
    if cubemap.dim() != 4 or cubemap.shape[3] != 3 or not MERGED_APPLY:
        return False
    n = 0
    r = R
    while r > min_resolution:
        if not tiles_eligible(r) or r % 2:
            return False
        r //= 2
        n += 1
    return n >= 2 and n <= 5 and tiles_eligible(r)


def _as_splitsum_fused(cubemap: Tensor, cutoff: float, min_resolution: int, min_roughness: float, max_roughness: float):
    mips = mip_chain(cubemap, min_resolution)
    n = len(mips)
    # (round 6, measured and removed: the diffuse map on a stream of its own beside the levels' launch -- a fifth stream shares one of
    #  HIP's four hardware queues with a front stream: 649 against 710 views/s)
    with torch.no_grad():
        base = diffuse_cubemap(mips[-1])
    roughs = _level_roughness(n, min_roughness, max_roughness)
    sizes = [m.numel() for m in mips]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=cubemap.device)
    levels = [t.view_as(m) for t, m in zip(torch.split(flat, sizes), mips)]
    jobs = []
    for m, out, rough in zip(mips, levels, roughs):
        e = specular_tiles(int(m.shape[1]), rough, cutoff, cubemap.device)
        jobs.append((e, m, out, 0, None))
    _tiles_apply_multi(jobs, "fwd")
    return base, levels


def as_splitsum(cubemap: Tensor, *, cutoff: float = 0.99, min_resolution: int = 16, min_roughness: float = 0.08,
                max_roughness: float = 0.5) -> TextureSplitSum:
    """TextureCubeMap.as_splitsum (rfstudio/graphics/_mesh/_texture.py:530-557); differentiable w.r.t. cubemap."""
    L.require_cuda(cubemap)
    if _fused_eligible(cubemap, min_resolution):
        out = _SplitSumFused.apply(cubemap.float(), float(cutoff), int(min_resolution), float(min_roughness), float(max_roughness))
        return TextureSplitSum(out[0], list(out[1:]), min_roughness, max_roughness)
    mips = [cubemap.float()]
    while mips[-1].shape[1] > min_resolution:
        mips.append(_CubeMapMip.apply(mips[-1]))
    assert len(mips) > 2, "Min resolution is too large."
    base = diffuse_cubemap(mips[-1])
    n = len(mips)
    levels = []
    for idx in range(n - 1):
        roughness = (idx / (n - 2)) * (max_roughness - min_roughness) + min_roughness
        levels.append(specular_cubemap(mips[idx], roughness, cutoff))
    levels.append(specular_cubemap(mips[-1], 1.0, cutoff))
    return TextureSplitSum(base, levels, min_roughness, max_roughness)


# ----------------------------------------------------------------------------- sharded prefilter (multi-GPU)
# The reference prefilters the environment once per step on its one device (rfstudio/model/geosplat.py:780-785); with one
# view per GPU the replicated prefilter would sit beside a ~2 ms view on every rank (Amdahl).  The tiled operator is independent
# per tile of output texels in both directions, so rank r applies its share of every tiled level's tile list (shard_tiles) into
# a zero-filled level and the ranks SUM (disjoint supports: x + 0, exact and identical everywhere):
#   forward : one all-reduce over the flat pyramid (25 MB for a 512^2 map)
#   backward: all-reduce of the 25 MB texel gradients (they are sums over the views of ALL ranks), then one all-reduce of the
#             per-level cubemap-gradient pieces; the mip chain / diffuse backward and the levels below 64^2 (direct kernels,
#             tens of microseconds) are replicated, so every rank ends with the same cubemap gradient and NO all-reduce of it
#             is needed.
def _level_roughness(n: int, min_roughness: float, max_roughness: float) -> List[float]:
    return [(idx / (n - 2)) * (max_roughness - min_roughness) + min_roughness for idx in range(n - 1)] + [1.0]


def can_shard_prefilter(cubemap_res: int, world: int, min_resolution: int = 16) -> bool:
    from .parallel import collectives_active
    return (world > 1 or collectives_active()) and tiles_eligible(cubemap_res) and cubemap_res >= 4 * min_resolution


def _flat_levels(res_list: List[int], device) -> Tuple[Tensor, List[Tensor]]:
    sizes = [6 * r * r * 3 for r in res_list]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
    return flat, [q.view(6, r, r, 3) for q, r in zip(torch.split(flat, sizes), res_list)]


def as_splitsum_sharded(cubemap: Tensor, rank: int, world: int, group=None, *, cutoff: float = 0.99, min_resolution: int = 16,
                        min_roughness: float = 0.08, max_roughness: float = 0.5) -> TextureSplitSum:
    """`as_splitsum` (no autograd graph) with the tiled levels computed 1/world per rank and summed."""
    import torch.distributed as dist
    L.require_cuda(cubemap)
    with torch.no_grad():
        mips = mip_chain(cubemap, min_resolution)
        assert len(mips) > 2, "Min resolution is too large."
        base = diffuse_cubemap(mips[-1])
        roughs = _level_roughness(len(mips), min_roughness, max_roughness)
        tiled = [i for i, m in enumerate(mips) if tiles_eligible(m.shape[1])]
        flat, parts = _flat_levels([mips[i].shape[1] for i in tiled], cubemap.device)
        levels: List[Optional[Tensor]] = [None] * len(mips)
        jobs = []
        for out, i in zip(parts, tiled):
            e = specular_tiles(mips[i].shape[1], roughs[i], cutoff, cubemap.device)
            t0, t1 = shard_tiles(e["n_tiles"], rank, world)
            jobs.append((e, mips[i], out, t0, t1))
            levels[i] = out
        _tiles_apply_multi(jobs, "fwd", world)
        if tiled:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        for i, m in enumerate(mips):
            if levels[i] is None:
                levels[i] = specular_cubemap(m, roughs[i], cutoff)          # small levels: replicated
    return TextureSplitSum(base, levels, min_roughness, max_roughness)


def as_splitsum_backward_sharded(g_base: Tensor, g_levels: List[Tensor], rank: int, world: int, group=None, *, cutoff: float = 0.99,
                                 min_roughness: float = 0.08, max_roughness: float = 0.5) -> Tensor:
    """Cubemap gradient from texel gradients that are ALREADY summed over the ranks; identical result on every rank."""
    import torch.distributed as dist
    n = len(g_levels)
    roughs = _level_roughness(n, min_roughness, max_roughness)
    tiled = [i for i, gl in enumerate(g_levels) if tiles_eligible(gl.shape[1])]
    flat, parts = _flat_levels([g_levels[i].shape[1] for i in tiled], g_base.device)
    g_mips: List[Optional[Tensor]] = [None] * n
    jobs = []
    for out, i in zip(parts, tiled):
        e = specular_tiles(g_levels[i].shape[1], roughs[i], cutoff, g_base.device)
        t0, t1 = shard_tiles(e["n_tiles"], rank, world)
        jobs.append((e, g_levels[i], out, t0, t1))
        g_mips[i] = out
    _tiles_apply_multi(jobs, "bwd", world)
    if tiled:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    for i in range(n):
        if g_mips[i] is None:
            g_mips[i] = _specular_level_backward(g_levels[i], roughs[i], cutoff)
    return _mip_chain_backward(g_mips, g_base)
