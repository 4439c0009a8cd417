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
        L.check(L.lib().gs_specular_bounds(res, L.f32(ct), L.ptr(b), L.stream()), "gs_specular_bounds")
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


_weights_cache: Dict[Tuple[int, float, float, int], Dict[str, Tensor]] = {}
CACHE_PAIR_WEIGHTS = os.environ.get("GEOSPLAT_PREFILTER_CACHE", "1") != "0"


def specular_weights(res: int, roughness: float, cutoff: float, device: torch.device) -> Dict[str, Tensor]:
    """Cached pair weights of one pyramid level (forward + transposed orientation, ~10 GB in total for a 512^2
    pyramid -- sized for the 288 GB of an MI355X).  They depend on (res, roughness, cutoff) only, while the cubemap
    changes every training step."""
    key = (res, float(roughness), float(cutoff), device.index or 0)
    if key not in _weights_cache:
        lib = L.lib()
        ct, bounds = specular_bounds(res, roughness, cutoff, device)
        table = dir_table(res, device)
        n = 6 * res * res
        counts = torch.empty(n, dtype=torch.int32, device=device)
        L.check(lib.gs_specular_patch_count(res, L.ptr(bounds), L.ptr(counts), L.stream()), "gs_specular_patch_count")
        csum = torch.cumsum(counts.long(), 0)
        offsets = (csum - counts.long()).contiguous()
        total = int(csum[-1].item())
        desc = torch.empty(max(total, 1), dtype=torch.int32, device=device)
        entry = {"offsets": offsets, "ct": ct, "bounds": bounds, "desc": desc, "total": total}
        for name, bwd in (("fwd", 0), ("bwd", 1)):
            w = torch.empty(max(total, 1) * 64, dtype=torch.float32, device=device)
            wsum = torch.empty(n, dtype=torch.float32, device=device) if not bwd else None
            L.check(lib.gs_specular_weights_build(res, L.ptr(bounds), L.ptr(table), L.ptr(offsets), L.f32(roughness), L.f32(ct),
                                                  bwd, L.ptr(w), L.ptr(wsum), L.ptr(desc), L.stream()),
                    "gs_specular_weights_build")
            entry[name] = w
            if wsum is not None:
                entry["wsum"] = wsum.view(6, res, res, 1)
        if DROP_EMPTY_PATCHES and total > 0:
            # 8x8 patches tile each face AABB of the lobe; the ones in the AABB corners hold no texel inside the lobe.
            # Membership (dot >= cutoff) is the same in both orientations, so one mask compacts fwd, bwd and the descriptors.
            keep = (entry["fwd"].view(total, 64) != 0).any(dim=1)
            csum_k = torch.cumsum(keep.long(), 0)
            new_total = int(csum_k[-1].item())
            before = torch.cat((csum_k.new_zeros(1), csum_k))             # kept patches in front of old patch p
            entry["offsets"] = before[offsets].contiguous()
            for name in ("fwd", "bwd"):
                entry[name] = entry[name].view(total, 64)[keep].reshape(-1).contiguous()
            entry["desc"] = desc[:total][keep].contiguous()
            entry["total"] = new_total
            entry["dropped"] = total - new_total
        _weights_cache[key] = entry
    return _weights_cache[key]


# ----------------------------------------------------------------------------- autograd pieces
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


DROP_EMPTY_PATCHES = os.environ.get("GEOSPLAT_DROP_EMPTY_PATCHES", "1") != "0"
APPLY_PACKED_SRC = os.environ.get("GEOSPLAT_APPLY_SRC", "3") == "3"   # 3: taps read the packed [6,R,R,3] map (one 12-byte load), 4: float4-padded copy


def _apply_src(t: Tensor):
    """(tensor, stride) of the tap source for gs_specular_apply"""
    if APPLY_PACKED_SRC:
        return t.contiguous(), 3
    return torch.nn.functional.pad(t, (0, 1)).contiguous(), 4


class _SpecularCubemapCached(torch.autograd.Function):
    """Same operator through the cached pair-weight tables (bit-identical weights, streamed instead of recomputed)."""

    @staticmethod
    def forward(ctx, cubemap: Tensor, res: int, roughness: float, cutoff: float) -> Tensor:
        e = specular_weights(res, roughness, cutoff, cubemap.device)
        src4, stride = _apply_src(cubemap)
        rgb = torch.empty(6, res, res, 3, dtype=torch.float32, device=cubemap.device)
        L.check(L.lib().gs_specular_apply(res, L.ptr(src4), stride, L.ptr(e["offsets"]), L.i64(e["total"]), L.ptr(e["desc"]),
                                          L.ptr(e["fwd"]), L.ptr(rgb), 3, 0, L.stream()), "gs_specular_apply")
        ctx.cfg = (res, roughness, cutoff)
        return rgb / e["wsum"]

    @staticmethod
    def backward(ctx, dout: Tensor):
        res, roughness, cutoff = ctx.cfg
        e = specular_weights(res, roughness, cutoff, dout.device)
        v4, stride = _apply_src(dout / e["wsum"])
        g = torch.empty_like(dout, memory_format=torch.contiguous_format)
        L.check(L.lib().gs_specular_apply(res, L.ptr(v4), stride, L.ptr(e["offsets"]), L.i64(e["total"]), L.ptr(e["desc"]),
                                          L.ptr(e["bwd"]), L.ptr(g), 3, 0, L.stream()), "gs_specular_apply")
        return g, None, None, None


def diffuse_cubemap(cubemap: Tensor) -> Tensor:
    return _DiffuseCubemap.apply(cubemap)


def specular_cubemap(cubemap: Tensor, roughness: float, cutoff: float = 0.99, cached: Optional[bool] = None) -> Tensor:
    """specular_cubemap of _wrap.py:138-157.  cached=True streams the pair weights from the per-level table
    (default, GEOSPLAT_PREFILTER_CACHE=0 disables), cached=False recomputes them in the kernel."""
    use_cache = CACHE_PAIR_WEIGHTS if cached is None else cached
    if use_cache:
        return _SpecularCubemapCached.apply(cubemap, int(cubemap.shape[1]), float(roughness), float(cutoff))
    ct, bounds = specular_bounds(cubemap.shape[1], roughness, cutoff, cubemap.device)
    return _SpecularCubemap.apply(cubemap, float(roughness), ct, bounds)


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


def as_splitsum_backward(g_base: Tensor, g_levels: List[Tensor], *, cutoff: float = 0.99, min_roughness: float = 0.08,
                         max_roughness: float = 0.5) -> Tensor:
    """Explicit backward of `as_splitsum` on the CURRENT stream (no autograd graph, so a caller can place it on any HIP
    stream -- the autograd engine would run it on the stream of the forward): texel gradients of the base map and of
    the n levels -> gradient of the cubemap.  Same kernels as the autograd path."""
    n = len(g_levels)
    roughs = [(idx / (n - 2)) * (max_roughness - min_roughness) + min_roughness for idx in range(n - 1)] + [1.0]
    g_mips = []
    for gl, rough in zip(g_levels, roughs):
        res = gl.shape[1]
        if CACHE_PAIR_WEIGHTS:
            e = specular_weights(res, rough, cutoff, gl.device)
            v4, stride = _apply_src(gl / e["wsum"])
            g = torch.empty(6, res, res, 3, dtype=torch.float32, device=gl.device)
            L.check(L.lib().gs_specular_apply(res, L.ptr(v4), stride, L.ptr(e["offsets"]), L.i64(e["total"]), L.ptr(e["desc"]),
                                              L.ptr(e["bwd"]), L.ptr(g), 3, 0, L.stream()), "gs_specular_apply")
        else:
            raise L.GeoSplatHipError("as_splitsum_backward needs the cached pair weights (GEOSPLAT_PREFILTER_CACHE=1)")
        g_mips.append(g)
    gd = g_base.contiguous()
    gdb = torch.empty_like(gd)
    L.check(L.lib().gs_diffuse_cubemap_bwd(gd.shape[1], L.ptr(gd), L.ptr(gdb), 0, L.stream()), "gs_diffuse_cubemap_bwd")
    g_mips[-1] = g_mips[-1] + gdb
    for idx in range(n - 1, 0, -1):
        dout = g_mips[idx].contiguous()
        R = dout.shape[1]
        up = torch.empty(6, 2 * R, 2 * R, 3, dtype=torch.float32, device=dout.device)
        L.check(L.lib().gs_cubemap_mip_bwd(R, L.ptr(dout), L.ptr(up), 0, L.stream()), "gs_cubemap_mip_bwd")
        g_mips[idx - 1] = g_mips[idx - 1] + up
    return g_mips[0]


def as_splitsum(cubemap: Tensor, *, cutoff: float = 0.99, min_resolution: int = 16, min_roughness: float = 0.08,
                max_roughness: float = 0.5) -> TextureSplitSum:
    """TextureCubeMap.as_splitsum (rfstudio/graphics/_mesh/_texture.py:530-557); differentiable w.r.t. cubemap."""
    L.require_cuda(cubemap)
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
# view per GPU that replicated 3.9 ms would sit beside a ~2.5 ms view on every rank (Amdahl).  The operator is independent
# per OUTPUT texel in both directions (forward: a level texel gathers cubemap texels; backward: a cubemap texel gathers
# level-gradient texels through the transposed tables), so rank r applies texels [r n/G, (r+1) n/G) of every level and the
# ranks all-gather in place:
#   forward : G x (1/G of the 12 GB weight stream)  + all-gather of the 25 MB pyramid
#   backward: all-reduce of the 25 MB texel gradients (they are sums over the views of ALL ranks), 1/G of the transposed
#             stream per rank, all-gather of the per-level cubemap-gradient pieces; the mip chain / diffuse backward are
#             cheap and replicated, so every rank ends with the same cubemap gradient and NO all-reduce of it is needed.
def _level_roughness(n: int, min_roughness: float, max_roughness: float) -> List[float]:
    return [(idx / (n - 2)) * (max_roughness - min_roughness) + min_roughness for idx in range(n - 1)] + [1.0]


def shard_texels(n_texels: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous share of a level's 6 R^2 texels; equal sizes (6 * 16^2 = 1536 divides by 2, 3, 4, 6, 8, ...)."""
    if n_texels % world != 0:
        raise L.GeoSplatHipError(f"{n_texels} texels do not split evenly over {world} ranks")
    per = n_texels // world
    return rank * per, (rank + 1) * per


def can_shard_prefilter(cubemap_res: int, world: int, min_resolution: int = 16) -> bool:
    return world > 1 and CACHE_PAIR_WEIGHTS and (6 * min_resolution * min_resolution) % world == 0 and cubemap_res >= 4 * min_resolution


def _all_gather_inplace(full: Tensor, t0: int, t1: int, group) -> None:
    """full: [n, 3] contiguous; rows [t0, t1) hold this rank's share -> every rank's share lands in place."""
    import torch.distributed as dist
    flat = full.view(-1)
    mine = flat[t0 * 3:t1 * 3]
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(flat, mine, group=group)             # in place: input is the rank-th slice of the output
    else:
        world = dist.get_world_size(group)
        per = (t1 - t0) * 3
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine.clone(), group=group)
        for r, part in enumerate(parts):
            flat[r * per:(r + 1) * per].copy_(part)


def _apply_range(e, table: str, src: Tensor, dst: Tensor, t0: int, t1: int, res: int) -> None:
    src_t, stride = _apply_src(src)
    L.check(L.lib().gs_specular_apply_range(res, L.ptr(src_t), stride, L.ptr(e["offsets"]), L.i64(e["total"]), L.ptr(e["desc"]),
                                            L.ptr(e[table]), L.ptr(dst), 3, 0, t0, t1, L.stream()), "gs_specular_apply_range")


def as_splitsum_sharded(cubemap: Tensor, rank: int, world: int, group=None, *, cutoff: float = 0.99, min_resolution: int = 16,
                        min_roughness: float = 0.08, max_roughness: float = 0.5) -> TextureSplitSum:
    """`as_splitsum` (no autograd graph) with the specular levels computed 1/world per rank and all-gathered."""
    L.require_cuda(cubemap)
    with torch.no_grad():
        mips = [cubemap.detach().float().contiguous()]
        while mips[-1].shape[1] > min_resolution:
            mips.append(_CubeMapMip.apply(mips[-1]))
        assert len(mips) > 2, "Min resolution is too large."
        base = diffuse_cubemap(mips[-1])
        levels = []
        for mip, rough in zip(mips, _level_roughness(len(mips), min_roughness, max_roughness)):
            res = mip.shape[1]
            e = specular_weights(res, rough, cutoff, mip.device)
            t0, t1 = shard_texels(6 * res * res, rank, world)
            out = torch.empty(6, res, res, 3, dtype=torch.float32, device=mip.device)
            _apply_range(e, "fwd", mip, out, t0, t1, res)
            o2 = out.view(-1, 3)
            o2[t0:t1].div_(e["wsum"].view(-1, 1)[t0:t1])
            _all_gather_inplace(o2, t0, t1, group)
            levels.append(out)
    return TextureSplitSum(base, levels, min_roughness, max_roughness)


def as_splitsum_backward_sharded(g_base: Tensor, g_levels: List[Tensor], rank: int, world: int, group=None, *, cutoff: float = 0.99,
                                 min_roughness: float = 0.08, max_roughness: float = 0.5) -> Tensor:
    """Cubemap gradient from texel gradients that are ALREADY summed over the ranks; identical result on every rank."""
    n = len(g_levels)
    g_mips = []
    for gl, rough in zip(g_levels, _level_roughness(n, min_roughness, max_roughness)):
        res = gl.shape[1]
        e = specular_weights(res, rough, cutoff, gl.device)
        t0, t1 = shard_texels(6 * res * res, rank, world)
        g = torch.empty(6, res, res, 3, dtype=torch.float32, device=gl.device)
        _apply_range(e, "bwd", gl / e["wsum"], g, t0, t1, res)
        _all_gather_inplace(g.view(-1, 3), t0, t1, group)
        g_mips.append(g)
    gd = g_base.contiguous()
    gdb = torch.empty_like(gd)
    L.check(L.lib().gs_diffuse_cubemap_bwd(gd.shape[1], L.ptr(gd), L.ptr(gdb), 0, L.stream()), "gs_diffuse_cubemap_bwd")
    g_mips[-1] = g_mips[-1] + gdb
    for idx in range(n - 1, 0, -1):
        dout = g_mips[idx].contiguous()
        R = dout.shape[1]
        up = torch.empty(6, 2 * R, 2 * R, 3, dtype=torch.float32, device=dout.device)
        L.check(L.lib().gs_cubemap_mip_bwd(R, L.ptr(dout), L.ptr(up), 0, L.stream()), "gs_cubemap_mip_bwd")
        g_mips[idx - 1] = g_mips[idx - 1] + up
    return g_mips[0]
