"""Second, independent CPU restatement (PyTorch autograd, any dtype) -- test infrastructure only.

Purpose: cross-check the hand-derived backward passes of the C oracle
(``gs_oracle*.c``) against automatic differentiation of the same forward
semantics in float64.  It is NOT the oracle the HIP path is graded against and
is never imported by ``geosplatting_amd``.

Forward semantics restated: SURVEY.md section 8a rows A1..A5 (gsplat 1.4 as
called at rfstudio/model/gsplat.py:334-355), S1..S4
(rfstudio/model/geosplat.py:80-122,474-476; rfstudio/graphics/_mesh/_texture.py:571-613).
"""
from __future__ import annotations

from typing import List, Optional

import torch


# ----------------------------------------------------------------------------- A1
def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
    ], -1)
    return R.reshape(q.shape[:-1] + (3, 3))


def project(means, quats, scales, viewmat, K, W, H, eps2d=0.3):
    """Differentiable part of A1 for ALL Gaussians (no culling): returns means2d, depths, conics, comp."""
    R = viewmat[:3, :3]; t = viewmat[:3, 3]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    mc = means @ R.T + t
    Rq = quat_to_rotmat(quats)
    M = Rq * scales[:, None, :]
    cov = M @ M.transpose(-1, -2)
    covc = R @ cov @ R.T
    x, y, z = mc.unbind(-1)
    tan_fovx = 0.5 * W / fx; tan_fovy = 0.5 * H / fy
    lim_x_pos = (W - cx) / fx + 0.3 * tan_fovx; lim_x_neg = cx / fx + 0.3 * tan_fovx
    lim_y_pos = (H - cy) / fy + 0.3 * tan_fovy; lim_y_neg = cy / fy + 0.3 * tan_fovy
    rz = 1.0 / z; rz2 = rz * rz
    tx = z * torch.minimum(lim_x_pos, torch.maximum(-lim_x_neg, x * rz))
    ty = z * torch.minimum(lim_y_pos, torch.maximum(-lim_y_neg, y * rz))
    zero = torch.zeros_like(z)
    J = torch.stack([fx * rz, zero, -fx * tx * rz2, zero, fy * rz, -fy * ty * rz2], -1).reshape(-1, 2, 3)
    cov2d = J @ covc @ J.transpose(-1, -2)
    det_orig = cov2d[:, 0, 0] * cov2d[:, 1, 1] - cov2d[:, 0, 1] * cov2d[:, 1, 0]
    c00 = cov2d[:, 0, 0] + eps2d; c11 = cov2d[:, 1, 1] + eps2d; c01 = cov2d[:, 0, 1]; c10 = cov2d[:, 1, 0]
    det = c00 * c11 - c01 * c10
    comp = torch.sqrt(torch.clamp(det_orig / det, min=0.0))
    conics = torch.stack([c11 / det, -c01 / det, c00 / det], -1)
    means2d = torch.stack([fx * x * rz + cx, fy * y * rz + cy], -1)
    return means2d, z, conics, comp


# ----------------------------------------------------------------------------- A5
def rasterize(means2d, conics, opacities, colors, W, H, tile_size, offsets, flatten_ids, n_isects=None):
    """Per-tile vectorised front-to-back compositing (same semantics as A5); differentiable."""
    tw = (W + tile_size - 1) // tile_size; th = (H + tile_size - 1) // tile_size
    D = colors.shape[1]
    dt = means2d.dtype
    render = torch.zeros(H, W, D, dtype=dt); alpha_img = torch.zeros(H, W, dtype=dt)
    off = offsets.reshape(-1).tolist()
    n_isects = flatten_ids.shape[0] if n_isects is None else n_isects
    for tile in range(tw * th):
        s = off[tile]; e = n_isects if tile == tw * th - 1 else off[tile + 1]
        ty, tx = divmod(tile, tw)
        y0, y1 = ty * tile_size, min((ty + 1) * tile_size, H)
        x0, x1 = tx * tile_size, min((tx + 1) * tile_size, W)
        if e <= s:
            continue
        g = flatten_ids[s:e].long()
        py, px = torch.meshgrid(torch.arange(y0, y1, dtype=dt) + 0.5, torch.arange(x0, x1, dtype=dt) + 0.5, indexing="ij")
        px = px.reshape(-1, 1); py = py.reshape(-1, 1)
        dx = means2d[g, 0][None] - px; dy = means2d[g, 1][None] - py
        a, b, c = conics[g, 0][None], conics[g, 1][None], conics[g, 2][None]
        sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy
        alpha = torch.clamp(opacities[g][None] * torch.exp(-sigma), max=0.999)
        valid = (sigma >= 0) & (alpha >= 1.0 / 255.0)
        alpha = torch.where(valid, alpha, torch.zeros_like(alpha))
        one_m = 1 - alpha
        T_incl = torch.cumprod(one_m, dim=1)
        T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], 1)
        stop = valid & (T_incl <= 1e-4)
        stopped = torch.cumsum(stop.to(torch.int64), 1) > 0          # includes the terminating Gaussian
        wgt = torch.where(stopped, torch.zeros_like(alpha), alpha * T_excl)
        out = wgt @ colors[g]
        # final transmittance = product over composited Gaussians only
        T_final = torch.prod(torch.where(stopped, torch.ones_like(one_m), one_m), dim=1)
        render[y0:y1, x0:x1] = out.reshape(y1 - y0, x1 - x0, D)
        alpha_img[y0:y1, x0:x1] = (1 - T_final).reshape(y1 - y0, x1 - x0)
    return render, alpha_img


# ----------------------------------------------------------------------------- S2/S3
def tex2d_linear_clamp(lut: torch.Tensor, uv: torch.Tensor) -> torch.Tensor:
    H, W, C = lut.shape
    x = (uv[:, 0] * W - 0.5).clamp(0, W - 1); y = (uv[:, 1] * H - 0.5).clamp(0, H - 1)
    ix0 = x.detach().floor().long(); iy0 = y.detach().floor().long()
    fx = (x - ix0)[:, None]; fy = (y - iy0)[:, None]
    ix1 = (ix0 + 1).clamp(max=W - 1); iy1 = (iy0 + 1).clamp(max=H - 1)
    top = lut[iy0, ix0] * (1 - fx) + lut[iy0, ix1] * fx
    bot = lut[iy1, ix0] * (1 - fx) + lut[iy1, ix1] * fx
    return top * (1 - fy) + bot * fy


_FACE = [  # (a, b, c, sx, sy): x = sx*d[a]/|d[c]|, y = sy*d[b]/|d[c]|
    (2, 1, 0, -1.0, -1.0), (2, 1, 0, 1.0, -1.0), (0, 2, 1, 1.0, 1.0),
    (0, 2, 1, 1.0, -1.0), (0, 1, 2, 1.0, -1.0), (0, 1, 2, -1.0, -1.0),
]


def _select_face(d: torch.Tensor) -> torch.Tensor:
    ax, ay, az = d.abs().unbind(-1)
    f = torch.where(az > torch.maximum(ax, ay), 4, torch.where(ay > ax, 2, 0))
    c = torch.where(f == 4, d[:, 2], torch.where(f == 2, d[:, 1], d[:, 0]))
    return f + (c < 0).long()


def _face_coords(d, face):
    x = torch.zeros_like(d[:, 0]); y = torch.zeros_like(d[:, 0])
    for s, (a, b, c, sx, sy) in enumerate(_FACE):
        m = face == s
        inv = 1.0 / d[:, c].abs().clamp(min=1e-30)
        x = torch.where(m, sx * d[:, a] * inv, x); y = torch.where(m, sy * d[:, b] * inv, y)
    return x, y


def _face_point(face, x, y):
    one = torch.ones_like(x)
    tabs = [(one, -y, -x), (-one, -y, x), (x, one, y), (x, -one, -y), (x, -y, one), (-x, -y, -one)]
    p = torch.zeros(x.shape[0], 3, dtype=x.dtype)
    for s, t in enumerate(tabs):
        m = (face == s)[:, None]
        p = torch.where(m, torch.stack(t, -1), p)
    return p


def _resolve(face, ix, iy, R, dt):
    ox = (ix < 0) | (ix >= R); oy = (iy < 0) | (iy >= R)
    xn = 2.0 * ((ix.to(dt) + 0.5) / R) - 1.0; yn = 2.0 * ((iy.to(dt) + 0.5) / R) - 1.0
    p = _face_point(face, xn, yn)
    f2 = _select_face(p)
    x2, y2 = _face_coords(p, f2)
    jx = ((x2 + 1) * 0.5 * R - 0.5 + 0.5).floor().long().clamp(0, R - 1)
    jy = ((y2 + 1) * 0.5 * R - 0.5 + 0.5).floor().long().clamp(0, R - 1)
    inside = ~(ox | oy)
    lin = torch.where(inside, (face * R + iy) * R + ix, (f2 * R + jy) * R + jx)
    corner = ox & oy
    return torch.where(corner, torch.full_like(lin, -1), lin)


def cube_linear(tex: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """tex [6,R,R,3], d [n,3] -> [n,3]; seam-aware bilinear as documented in gs_oracle_shade.c"""
    R = tex.shape[1]; dt = d.dtype
    face = _select_face(d.detach())
    x, y = _face_coords(d, face)
    tx = (x + 1) * 0.5 * R - 0.5; ty = (y + 1) * 0.5 * R - 0.5
    ix0 = tx.detach().floor().long(); iy0 = ty.detach().floor().long()
    fx = (tx - ix0)[:, None]; fy = (ty - iy0)[:, None]
    flat = tex.reshape(-1, 3)
    idx = [_resolve(face, ix0 + ox, iy0 + oy, R, dt) for (ox, oy) in ((0, 0), (1, 0), (0, 1), (1, 1))]
    vals = [flat[i.clamp(min=0)] for i in idx]
    miss = [(i < 0)[:, None] for i in idx]
    tot = sum(torch.where(m, torch.zeros_like(v), v) for v, m in zip(vals, miss))
    vals = [torch.where(m, tot / 3.0, v) for v, m in zip(vals, miss)]
    top = vals[0] + fx * (vals[1] - vals[0]); bot = vals[2] + fx * (vals[3] - vals[2])
    return top + fy * (bot - top)


def mip_from_roughness(r, min_r=0.08, max_r=0.5, L=6):
    lo = ((r - min_r) / (max_r - min_r)).clamp(0, 1) * (L - 2)
    hi = ((r - max_r) / (1.0 - max_r)).clamp(0, 1) + (L - 2)
    return torch.where(r < max_r, lo, hi)


def cube_mip_linear(levels: List[torch.Tensor], d: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    L = len(levels)
    lam = bias.clamp(0, L - 1)
    l0 = lam.detach().floor().long().clamp(max=L - 1)
    f = (lam - l0)[:, None]
    out = torch.zeros(d.shape[0], 3, dtype=d.dtype)
    for l in range(L):
        m0 = l0 == l
        if m0.any():
            c0 = cube_linear(levels[l], d[m0])
            if l < L - 1:
                c1 = cube_linear(levels[l + 1], d[m0])
                c0 = c0 + f[m0] * (c1 - c0)
            out = out.index_put((m0.nonzero()[:, 0],), c0)
    return out


# ----------------------------------------------------------------------------- S1
def safe_normalize(v):
    l = v.norm(dim=-1, keepdim=True)
    return torch.where(l < 1e-6, torch.tensor([0.0, 0.0, 1.0], dtype=v.dtype), v / l.clamp_min(1e-6))


def shade(means, normals, kd, ks, cam_pos, lut, base, levels, min_roughness=0.1, max_metallic=1.0, mode="pbr",
          env_min_r=0.08, env_max_r=0.5):
    rough = ks[:, 0:1] * (1 - min_roughness) + min_roughness
    metal = ks[:, 1:2] * max_metallic
    spec = (1.0 - metal) * 0.04 + kd * metal
    diff = kd * (1.0 - metal)
    wo = safe_normalize(cam_pos - means)
    d = (normals * wo).sum(-1, keepdim=True)
    ndv = d.clamp(min=1e-6)
    fg = tex2d_linear_clamp(lut, torch.cat([ndv, rough], -1))
    refl = 2 * d * normals - wo
    l_diff = cube_linear(base, normals)
    l_spec = cube_mip_linear(levels, refl, mip_from_roughness(rough[:, 0], env_min_r, env_max_r, len(levels)))
    reflectance = spec * fg[:, 0:1] + fg[:, 1:2]
    if mode == "pbr":
        return diff + l_spec * reflectance
    if mode == "diffuse":
        return l_diff * diff
    return l_spec * reflectance


# ----------------------------------------------------------------------------- S4
def tonemap_naive(rgba, exposure):
    rgb = rgba[..., :3] * exposure
    return torch.cat((1 - torch.nn.functional.softplus(1 - rgb, beta=100), rgba[..., 3:]), -1)


def tonemap_aces(rgba, exposure):
    rgb = rgba[..., :3] * exposure
    return torch.cat(((rgb * (2.51 * rgb + 0.03)) / (rgb * (2.43 * rgb + 0.59) + 0.14), rgba[..., 3:]), -1)
