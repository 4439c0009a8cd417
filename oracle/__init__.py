"""CPU ORACLE -- test infrastructure, NOT product code.

ctypes/numpy front end of ``oracle/libgs_oracle.so`` (plain-C restatement of the
reference's render-and-backward path, see the headers of ``gs_oracle*.c`` for
the reference file:line each function follows).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``geosplatting_amd`` never does.

PARITY UNPINNED at the gsplat / nvdiffrast boundary (un-vendored third-party
CUDA packages, no golden vectors in the reference); the pure-Python glue of the
reference (camera matrices, S1 arithmetic, tone mapping, mip map, atlas packing)
is pinned by ``tests/golden/*.npz`` generated from the importable reference
modules by ``scripts/make_golden.py``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgs_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("gs_oracle.c", "gs_oracle_shade.c", "gs_oracle_splitsum.c")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgs_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gso_project_fwd.restype = C.c_int
        _lib.gso_isect_count.restype = C.c_int64
        _lib.gso_mip_from_roughness.restype = C.c_float
        _lib.gso_mip_from_roughness.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _fl(x: float) -> C.c_float:
    return C.c_float(float(x))


# ----------------------------------------------------------------------------- A1..A7
def project_fwd(means, quats, scales, viewmat, K, W, H, eps2d=0.3, near=0.01, far=1e10, radius_clip=0.0):
    means, quats, scales = _f32(means), _f32(quats), _f32(scales)
    viewmat, K = _f32(viewmat).reshape(4, 4), _f32(K).reshape(3, 3)
    N = means.shape[0]
    gids = np.empty(N, np.int32); radii = np.empty(N, np.int32)
    means2d = np.empty((N, 2), np.float32); depths = np.empty(N, np.float32)
    conics = np.empty((N, 3), np.float32); comps = np.empty(N, np.float32)
    V = lib().gso_project_fwd(N, _p(means), _p(quats), _p(scales), _p(viewmat), _p(K), int(W), int(H),
                              _fl(eps2d), _fl(near), _fl(far), _fl(radius_clip),
                              _p(gids), _p(radii), _p(means2d), _p(depths), _p(conics), _p(comps))
    return dict(gaussian_ids=gids[:V].copy(), radii=radii[:V].copy(), means2d=means2d[:V].copy(),
                depths=depths[:V].copy(), conics=conics[:V].copy(), compensations=comps[:V].copy())


def isect_tiles(means2d, radii, depths, tile_size, tile_w, tile_h):
    means2d, depths = _f32(means2d), _f32(depths)
    radii = np.ascontiguousarray(radii, np.int32)
    V = radii.shape[0]
    tpg = np.empty(V, np.int32)
    n = lib().gso_isect_count(V, _p(means2d), _p(radii), int(tile_size), int(tile_w), int(tile_h), _p(tpg))
    isect_ids = np.empty(n, np.int64); flatten_ids = np.empty(n, np.int32)
    lib().gso_isect_emit(V, _p(means2d), _p(radii), _p(depths), int(tile_size), int(tile_w), int(tile_h),
                         _p(isect_ids), _p(flatten_ids))
    return tpg, isect_ids, flatten_ids


def sort_pairs(isect_ids, flatten_ids):
    k = np.ascontiguousarray(isect_ids, np.int64).copy()
    v = np.ascontiguousarray(flatten_ids, np.int32).copy()
    lib().gso_sort_pairs(C.c_int64(k.shape[0]), _p(k), _p(v))
    return k, v


def isect_offsets(isect_ids_sorted, n_tiles):
    off = np.empty(n_tiles, np.int32)
    k = np.ascontiguousarray(isect_ids_sorted, np.int64)
    lib().gso_isect_offsets(C.c_int64(k.shape[0]), _p(k), int(n_tiles), _p(off))
    return off


def raster_fwd(W, H, tile_size, means2d, conics, opacities, colors, offsets, flatten_ids, background=None):
    means2d, conics, opacities, colors = _f32(means2d), _f32(conics), _f32(opacities), _f32(colors)
    D = colors.shape[1]
    assert D <= 64
    offsets = np.ascontiguousarray(offsets, np.int32); flatten_ids = np.ascontiguousarray(flatten_ids, np.int32)
    render = np.empty((H, W, D), np.float32); alphas = np.empty((H, W), np.float32)
    last_ids = np.empty((H, W), np.int32); amb = np.empty((H, W), np.uint8)
    pairs = (C.c_int64 * 2)(0, 0)          # evaluated pairs, composited (valid) pairs
    bg = None if background is None else _f32(background)
    lib().gso_raster_fwd(int(W), int(H), int(tile_size), int(D), _p(means2d), _p(conics), _p(opacities), _p(colors),
                         _p(bg), C.c_int64(flatten_ids.shape[0]), _p(offsets), _p(flatten_ids),
                         _p(render), _p(alphas), _p(last_ids), _p(amb), pairs)
    return render, alphas, last_ids, amb.astype(bool), (int(pairs[0]), int(pairs[1]))


def raster_bwd(W, H, tile_size, means2d, conics, opacities, colors, offsets, flatten_ids, alphas, last_ids,
               v_render, v_alphas, background=None):
    means2d, conics, opacities, colors = _f32(means2d), _f32(conics), _f32(opacities), _f32(colors)
    V, D = colors.shape
    offsets = np.ascontiguousarray(offsets, np.int32); flatten_ids = np.ascontiguousarray(flatten_ids, np.int32)
    alphas = _f32(alphas); last_ids = np.ascontiguousarray(last_ids, np.int32)
    v_render, v_alphas = _f32(v_render), _f32(v_alphas)
    bg = None if background is None else _f32(background)
    v_means2d = np.empty((V, 2), np.float32); v_conics = np.empty((V, 3), np.float32)
    v_colors = np.empty((V, D), np.float32); v_opac = np.empty(V, np.float32)
    lib().gso_raster_bwd(int(W), int(H), int(tile_size), int(D), _p(means2d), _p(conics), _p(opacities), _p(colors),
                         _p(bg), C.c_int64(flatten_ids.shape[0]), _p(offsets), _p(flatten_ids), _p(alphas),
                         _p(last_ids), _p(v_render), _p(v_alphas), int(V),
                         _p(v_means2d), _p(v_conics), _p(v_colors), _p(v_opac))
    return v_means2d, v_conics, v_colors, v_opac


def project_bwd(means, quats, scales, opacities, viewmat, K, W, H, gaussian_ids, conics, compensations,
                v_means2d, v_conics, v_opacities_packed, v_colors_packed, v_depths=None, eps2d=0.3):
    means, quats, scales, opacities = _f32(means), _f32(quats), _f32(scales), _f32(opacities)
    viewmat, K = _f32(viewmat).reshape(4, 4), _f32(K).reshape(3, 3)
    gids = np.ascontiguousarray(gaussian_ids, np.int32)
    conics, comps = _f32(conics), _f32(compensations)
    v_means2d, v_conics = _f32(v_means2d), _f32(v_conics)
    v_op, v_col = _f32(v_opacities_packed), _f32(v_colors_packed)
    vd = None if v_depths is None else _f32(v_depths)
    N, V, D = means.shape[0], gids.shape[0], v_col.shape[1]
    g_means = np.empty((N, 3), np.float32); g_quats = np.empty((N, 4), np.float32)
    g_scales = np.empty((N, 3), np.float32); g_opac = np.empty(N, np.float32); g_colors = np.empty((N, D), np.float32)
    lib().gso_project_bwd(int(N), int(V), int(D), _p(means), _p(quats), _p(scales), _p(opacities), _p(viewmat), _p(K),
                          int(W), int(H), _fl(eps2d), _p(gids), _p(conics), _p(comps), _p(v_means2d), _p(vd),
                          _p(v_conics), _p(v_op), _p(v_col),
                          _p(g_means), _p(g_quats), _p(g_scales), _p(g_opac), _p(g_colors))
    return g_means, g_quats, g_scales, g_opac, g_colors


def rasterization(means, quats, scales, opacities, colors, viewmat, K, W, H, tile_size=16,
                  eps2d=0.3, near=0.01, far=1e10, radius_clip=0.0, background=None) -> Dict[str, np.ndarray]:
    """Whole forward of ``gsplat.rasterization`` as the reference calls it
    (rfstudio/model/gsplat.py:334-355): A1, A1', A2, A3, A4, A5."""
    opacities, colors = _f32(opacities), _f32(colors)
    m = project_fwd(means, quats, scales, viewmat, K, W, H, eps2d, near, far, radius_clip)
    gid = m["gaussian_ids"]
    m["opacities"] = (opacities[gid] * m["compensations"]).astype(np.float32)
    m["colors"] = colors[gid]
    tw, th = (W + tile_size - 1) // tile_size, (H + tile_size - 1) // tile_size
    tpg, ids, flat = isect_tiles(m["means2d"], m["radii"], m["depths"], tile_size, tw, th)
    ids, flat = sort_pairs(ids, flat)
    off = isect_offsets(ids, tw * th)
    render, alphas, last_ids, amb, pairs = raster_fwd(W, H, tile_size, m["means2d"], m["conics"], m["opacities"],
                                                       m["colors"], off, flat, background)
    m.update(tiles_per_gauss=tpg, isect_ids=ids, flatten_ids=flat, isect_offsets=off.reshape(th, tw),
             render=render, alphas=alphas, last_ids=last_ids, ambiguous=amb, pairs=pairs[0], pairs_valid=pairs[1],
             tile_width=tw, tile_height=th)
    return m


def rasterization_bwd(means, quats, scales, opacities, colors, viewmat, K, W, H, meta, v_render, v_alphas,
                      tile_size=16, eps2d=0.3, background=None):
    """Whole backward: A6 then A7 (+ gather backward)."""
    v_m2d, v_con, v_col, v_op = raster_bwd(W, H, tile_size, meta["means2d"], meta["conics"], meta["opacities"],
                                           meta["colors"], meta["isect_offsets"].reshape(-1), meta["flatten_ids"],
                                           meta["alphas"], meta["last_ids"], v_render, v_alphas, background)
    g = project_bwd(means, quats, scales, opacities, viewmat, K, W, H, meta["gaussian_ids"], meta["conics"],
                    meta["compensations"], v_m2d, v_con, v_op, v_col, eps2d=eps2d)
    return dict(v_means=g[0], v_quats=g[1], v_scales=g[2], v_opacities=g[3], v_colors=g[4],
                v_means2d=v_m2d, v_conics=v_con, v_colors_packed=v_col, v_opacities_packed=v_op)


# ----------------------------------------------------------------------------- S4
TONE = {"none": 0, "naive": 1, "aces": 2}


def tonemap_fwd(rgba, exposure, mode="naive"):
    rgba = _f32(rgba); out = np.empty_like(rgba)
    lib().gso_tonemap_fwd(C.c_int64(rgba.size // 4), TONE[mode], _p(rgba), _fl(exposure), _p(out))
    return out


def tonemap_bwd(rgba, exposure, v_out, mode="naive"):
    rgba, v_out = _f32(rgba), _f32(v_out)
    v_rgba = np.empty_like(rgba); ve = C.c_float(0)
    lib().gso_tonemap_bwd(C.c_int64(rgba.size // 4), TONE[mode], _p(rgba), _fl(exposure), _p(v_out), _p(v_rgba),
                          C.byref(ve))
    return v_rgba, float(ve.value)


# ----------------------------------------------------------------------------- S1..S3
MODE = {"pbr": 0, "diffuse": 1, "specular": 2}


def _ptr_array(arrs: Sequence[np.ndarray]):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def shade_fwd(means, normals, kd, ks, cam_pos, lut, base, levels: List[np.ndarray], min_roughness=0.1,
              max_metallic=1.0, mode="pbr", env_min_roughness=0.08, env_max_roughness=0.5):
    means, normals, kd, ks, cam_pos = _f32(means), _f32(normals), _f32(kd), _f32(ks), _f32(cam_pos)
    lut, base = _f32(lut), _f32(base)
    levels = [_f32(l) for l in levels]
    res = np.array([l.shape[1] for l in levels], np.int32)
    N = means.shape[0]
    colors = np.empty((N, 3), np.float32)
    lib().gso_shade_fwd(int(N), _p(means), _p(normals), _p(kd), _p(ks), _p(cam_pos), _fl(min_roughness),
                        _fl(max_metallic), MODE[mode], _p(lut), int(lut.shape[-2]), _p(base), int(base.shape[1]),
                        _ptr_array(levels), _p(res), len(levels), _fl(env_min_roughness), _fl(env_max_roughness),
                        _p(colors))
    return colors


def shade_bwd(means, normals, kd, ks, cam_pos, lut, base, levels: List[np.ndarray], v_colors, min_roughness=0.1,
              max_metallic=1.0, mode="pbr", env_min_roughness=0.08, env_max_roughness=0.5):
    means, normals, kd, ks, cam_pos = _f32(means), _f32(normals), _f32(kd), _f32(ks), _f32(cam_pos)
    lut, base, v_colors = _f32(lut), _f32(base), _f32(v_colors)
    levels = [_f32(l) for l in levels]
    res = np.array([l.shape[1] for l in levels], np.int32)
    N = means.shape[0]
    v_means = np.empty((N, 3), np.float32); v_normals = np.empty((N, 3), np.float32)
    v_kd = np.empty((N, 3), np.float32); v_ks = np.empty((N, 2), np.float32)
    v_base = np.zeros_like(base); v_levels = [np.zeros_like(l) for l in levels]
    lib().gso_shade_bwd(int(N), _p(means), _p(normals), _p(kd), _p(ks), _p(cam_pos), _fl(min_roughness),
                        _fl(max_metallic), MODE[mode], _p(lut), int(lut.shape[-2]), _p(base), int(base.shape[1]),
                        _ptr_array(levels), _p(res), len(levels), _fl(env_min_roughness), _fl(env_max_roughness),
                        _p(v_colors), _p(v_means), _p(v_normals), _p(v_kd), _p(v_ks), _p(v_base),
                        _ptr_array(v_levels))
    return dict(v_means=v_means, v_normals=v_normals, v_kd=v_kd, v_ks=v_ks, v_base=v_base, v_levels=v_levels)


def tex2d_linear_clamp(lut, uv):
    lut, uv = _f32(lut), _f32(uv)
    H, W, Cn = lut.shape[-3:]
    n = uv.shape[0]
    out = np.empty((n, Cn), np.float32); du = np.empty((n, Cn), np.float32); dv = np.empty((n, Cn), np.float32)
    lib().gso_tex2d_linear_clamp(int(n), _p(lut), int(W), int(H), int(Cn), _p(uv), _p(out), _p(du), _p(dv))
    return out, du, dv


def cube_linear(tex, dirs):
    tex, dirs = _f32(tex), _f32(dirs)
    n = dirs.shape[0]
    out = np.empty((n, 3), np.float32); dd = np.empty((n, 3, 3), np.float32)
    lib().gso_cube_linear(int(n), _p(tex), int(tex.shape[1]), _p(dirs), _p(out), _p(dd))
    return out, dd


def cube_mip_linear(levels, dirs, bias):
    levels = [_f32(l) for l in levels]; dirs = _f32(dirs); bias = _f32(bias)
    res = np.array([l.shape[1] for l in levels], np.int32)
    n = dirs.shape[0]
    out = np.empty((n, 3), np.float32); dd = np.empty((n, 3, 3), np.float32); dm = np.empty((n, 3), np.float32)
    lib().gso_cube_mip_linear(int(n), _ptr_array(levels), _p(res), len(levels), _p(dirs), _p(bias), _p(out), _p(dd),
                              _p(dm))
    return out, dd, dm


def mip_from_roughness(r, min_r=0.08, max_r=0.5, L=6) -> float:
    return float(lib().gso_mip_from_roughness(float(r), float(min_r), float(max_r), int(L)))


# ----------------------------------------------------------------------------- S5
def cubemap_mip_fwd(cubemap):
    cubemap = _f32(cubemap)
    R, Cn = cubemap.shape[1], cubemap.shape[3]
    out = np.empty((6, R // 2, R // 2, Cn), np.float32)
    lib().gso_cubemap_mip_fwd(int(R), int(Cn), _p(cubemap), _p(out))
    return out


def cubemap_mip_bwd(v_out):
    """_CubeMapMip.backward (rfstudio/graphics/_mesh/_texture.py:208-226): NOT the adjoint of the average --
    a seam-aware bilinear cube lookup of 0.25*dout at every fine texel's direction."""
    v_out = _f32(v_out)
    res = v_out.shape[1] * 2
    out = np.empty((6, res, res, 3), np.float32)
    g = np.linspace(-1.0 + 1.0 / res, 1.0 - 1.0 / res, res, dtype=np.float32)
    gy, gx = np.meshgrid(g, g, indexing="ij")
    one = np.ones_like(gx)
    tab = [(one, -gy, -gx), (-one, -gy, gx), (gx, one, gy), (gx, -one, -gy), (gx, -gy, one), (-gx, -gy, -one)]
    quarter = (v_out * np.float32(0.25)).astype(np.float32)
    for s in range(6):
        v = np.stack(tab[s], -1).astype(np.float32)
        v = v / np.linalg.norm(v, axis=-1, keepdims=True).astype(np.float32)
        out[s] = cube_linear(quarter, v.reshape(-1, 3))[0].reshape(res, res, 3)
    return out


def diffuse_cubemap_fwd(cubemap):
    cubemap = _f32(cubemap); out = np.empty_like(cubemap)
    lib().gso_diffuse_cubemap_fwd(int(cubemap.shape[1]), _p(cubemap), _p(out))
    return out


def diffuse_cubemap_bwd(v_out):
    v_out = _f32(v_out); g = np.empty_like(v_out)
    lib().gso_diffuse_cubemap_bwd(int(v_out.shape[1]), _p(v_out), _p(g))
    return g


def exp_neg_check(lo: float, hi: float) -> Tuple[float, int]:
    """(max relative error vs float64 exp, order-independent checksum of the result bits) of the canonical exp(-sigma) the
    compositor restatement uses (gs_oracle.c gso_exp_neg) over EVERY float in [lo, hi]."""
    lo_b = int(np.float32(lo).view(np.uint32)); hi_b = int(np.float32(hi).view(np.uint32))
    worst = C.c_double(0.0); cs = C.c_uint64(0)
    lib().gso_exp_neg_check(C.c_uint32(lo_b), C.c_uint32(hi_b), C.byref(worst), C.byref(cs))
    return float(worst.value), int(cs.value)


def ndf_cutoff(roughness: float, cutoff: float = 0.99, n_samples: int = 1000000) -> float:
    """cos(theta) that retains `cutoff` of the GGX NDF energy
    (rfstudio/graphics/_mesh/_splitsum/_wrap.py:120-135, float64 numpy like the reference)."""
    def ndf(alpha_sqr, costheta):
        costheta = np.clip(costheta, 0.0, 1.0)
        d = (costheta * alpha_sqr - costheta) * costheta + 1.0
        return alpha_sqr / (d * d * np.pi)
    costheta = np.cos(np.linspace(0, np.pi / 2.0, n_samples))
    D = np.cumsum(ndf(roughness ** 4, costheta))
    idx = np.argmax(D >= D[..., -1] * cutoff)
    return float(costheta[idx])


def specular_bounds(R: int, costheta_cutoff: float):
    b = np.empty((6, R, R, 24), np.float32)
    lib().gso_specular_bounds(int(R), _fl(costheta_cutoff), _p(b))
    return b


def specular_cubemap_fwd(cubemap, bounds, roughness, costheta_cutoff):
    cubemap, bounds = _f32(cubemap), _f32(bounds)
    R = cubemap.shape[1]
    out = np.empty((6, R, R, 4), np.float32)
    lib().gso_specular_cubemap_fwd(int(R), _p(cubemap), _p(bounds), _fl(roughness), _fl(costheta_cutoff), _p(out))
    return out


def specular_cubemap_bwd(bounds, v_out_rgb, roughness, costheta_cutoff):
    bounds, v = _f32(bounds), _f32(v_out_rgb)
    g = np.empty_like(v)
    lib().gso_specular_cubemap_bwd(int(v.shape[1]), _p(bounds), _p(v), _fl(roughness), _fl(costheta_cutoff), _p(g))
    return g


def specular_subset(cubemap, sel, roughness, costheta_cutoff):
    """Forward prefilter of the output texels `sel` (flat indices (s*R+y)*R+x) only: returns (out[n,4] = (sum rgb*w, sum w),
    bounds[n,24]).  Same statements as the full loops (parity tests at R = 512 / 256 / 128)."""
    cubemap = _f32(cubemap); R = cubemap.shape[1]
    sel = np.ascontiguousarray(sel, dtype=np.int32); n = int(sel.shape[0])
    b = np.empty((n, 24), np.float32); out = np.empty((n, 4), np.float32)
    lib().gso_specular_bounds_subset(int(R), _fl(costheta_cutoff), n, sel.ctypes.data_as(C.c_void_p), _p(b))
    lib().gso_specular_cubemap_fwd_subset(int(R), _p(cubemap), n, sel.ctypes.data_as(C.c_void_p), _p(b), _fl(roughness),
                                          _fl(costheta_cutoff), _p(out))
    return out, b


def specular_subset_bwd(R, sel, bounds, v_out_rgb, roughness, costheta_cutoff):
    """Cubemap gradient [6,R,R,3] of a cotangent that is non-zero on the texels `sel` only (v_out_rgb[n,3] w.r.t. the
    un-normalised rgb sums)."""
    sel = np.ascontiguousarray(sel, dtype=np.int32); n = int(sel.shape[0])
    bounds, v = _f32(bounds), _f32(v_out_rgb)
    g = np.empty((6, R, R, 3), np.float32)
    lib().gso_specular_cubemap_bwd_subset(int(R), n, sel.ctypes.data_as(C.c_void_p), _p(bounds), _p(v), _fl(roughness),
                                          _fl(costheta_cutoff), _p(g))
    return g


def splitsum_roughness(L, min_roughness=0.08, max_roughness=0.5):
    """per-level prefilter roughness (rfstudio/graphics/_mesh/_texture.py:545-548)"""
    return [(i / (L - 2)) * (max_roughness - min_roughness) + min_roughness for i in range(L - 1)] + [1.0]


def as_splitsum(cubemap, cutoff=0.99, min_resolution=16, min_roughness=0.08, max_roughness=0.5):
    """TextureCubeMap.as_splitsum (rfstudio/graphics/_mesh/_texture.py:530-557).
    Returns (base[6,16,16,3], levels list of [6,R_l,R_l,3], saved-state for the backward)."""
    mips = [_f32(cubemap)]
    while mips[-1].shape[1] > min_resolution:
        mips.append(cubemap_mip_fwd(mips[-1]))
    assert len(mips) > 2
    base = diffuse_cubemap_fwd(mips[-1])
    L = len(mips)
    rough = [(i / (L - 2)) * (max_roughness - min_roughness) + min_roughness for i in range(L - 1)] + [1.0]
    levels, saved = [], []
    for i in range(L):
        R = mips[i].shape[1]
        ct = ndf_cutoff(rough[i], cutoff)
        b = specular_bounds(R, ct)
        raw = specular_cubemap_fwd(mips[i], b, rough[i], ct)
        levels.append((raw[..., :3] / raw[..., 3:]).astype(np.float32))
        saved.append((b, rough[i], ct, raw[..., 3:].copy()))
    return base, levels, saved


def as_splitsum_bwd(saved, v_base, v_levels):
    """Gradient of as_splitsum w.r.t. the cubemap parameter (sum over the whole mip chain)."""
    L = len(v_levels)
    g_mips = []
    for i in range(L):
        b, r, ct, wsum = saved[i]
        g_mips.append(specular_cubemap_bwd(b, (_f32(v_levels[i]) / wsum).astype(np.float32), r, ct))
    g_mips[-1] = g_mips[-1] + diffuse_cubemap_bwd(v_base)
    for i in range(L - 1, 0, -1):
        g_mips[i - 1] = g_mips[i - 1] + cubemap_mip_bwd(g_mips[i])
    return g_mips[0]
