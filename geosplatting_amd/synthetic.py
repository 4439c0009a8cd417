"""Seeded synthetic inputs of the shapes BASELINE.json names (no datasets exist on either box).

  * ``random_splats``      : config 0 -- semantics of Splats.random (rfstudio/graphics/_splats.py:61-80)
  * ``icosphere`` + ``mesh_to_splats`` : configs 1..3 -- GeoSplatting-like surface splats: the reference's
    MGAdapter (6 flat Gaussians per face, rfstudio/model/geosplat.py:378-472) applied to a bumpy icosphere;
    level 6 -> 491 520 Gaussians, level 7 -> 1 966 080 Gaussians
  * ``blender_cameras``    : 800x800, fx=fy=0.5*800/tan(0.5*0.6911112), eye distance 4*(2/3)
                             (rfstudio/data/dataparser/syn4relight_dataparser.py:42-75)
  * ``make_cubemap``       : seeded smooth HDR environment >= 1e-2 (rfstudio/trainer/geosplat_trainer.py:266)

Everything here is host-side torch on CPU (moved to the GPU by the caller).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import torch
from torch import Tensor

from .cameras import Camera, orbit_cameras


# ----------------------------------------------------------------------------- geometry helpers
def safe_normalize(v: Tensor) -> Tensor:
    """rfstudio/graphics/math.py:119-125"""
    l = v.norm(dim=-1, keepdim=True)
    return torch.where(l < 1e-6, torch.tensor([0.0, 0.0, 1.0], dtype=v.dtype, device=v.device), v / l.clamp_min(1e-6))


def rot2quat(rots: Tensor) -> Tensor:
    """Rotation matrices [*,3,3] -> quaternions wxyz [*,4]; best-conditioned branch of the four
    candidates (semantics of rfstudio/graphics/math.py:246-278)."""
    m = rots.reshape(-1, 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    q = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)
    q_abs = torch.sqrt(q.clamp_min(0.0))
    cand = torch.stack([
        torch.stack([q_abs[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, q_abs[:, 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, q_abs[:, 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[:, 3] ** 2], -1),
    ], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp_min(0.1))
    best = q_abs.argmax(-1)
    out = cand[torch.arange(cand.shape[0]), best]
    return out.reshape(rots.shape[:-2] + (4,))


def icosphere(level: int, radius: float = 1.0) -> Tuple[Tensor, Tensor]:
    """Unit icosahedron subdivided `level` times -> (vertices [V,3] fp32, faces [F,3] int64), F = 20*4^level."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = torch.tensor([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                      [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=torch.float64)
    f = torch.tensor([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                      [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                      [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=torch.int64)
    v = v / v.norm(dim=-1, keepdim=True)
    for _ in range(level):
        nv = v.shape[0]
        e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
        e_sorted, _ = e.sort(dim=1)
        key = e_sorted[:, 0] * nv + e_sorted[:, 1]
        uniq, inv = torch.unique(key, return_inverse=True)
        a = torch.div(uniq, nv, rounding_mode="floor"); b = uniq % nv
        mid = v[a] + v[b]
        mid = mid / mid.norm(dim=-1, keepdim=True)
        v = torch.cat([v, mid], 0)
        F = f.shape[0]
        m01 = nv + inv[:F]; m12 = nv + inv[F:2 * F]; m20 = nv + inv[2 * F:]
        f = torch.cat([
            torch.stack([f[:, 0], m01, m20], 1), torch.stack([f[:, 1], m12, m01], 1),
            torch.stack([f[:, 2], m20, m12], 1), torch.stack([m01, m12, m20], 1)], 0)
    return (v * radius).float(), f


def vertex_normals(vertices: Tensor, faces: Tensor) -> Tensor:
    """Area-weighted vertex normals (rfstudio/graphics/_mesh/_triangle_mesh.py:588-615)."""
    p = vertices[faces]                                     # [F,3,3]
    fn = torch.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], dim=-1)
    n = torch.zeros_like(vertices)
    n.index_add_(0, faces.reshape(-1), fn[:, None, :].expand(-1, 3, -1).reshape(-1, 3))
    return n / n.norm(dim=-1, keepdim=True)


@dataclass
class SplatSet:
    """Raw (pre-activation) Gaussian parameters, field layout of rfstudio.graphics.Splats
    (rfstudio/graphics/_splats.py:18-32): log-scales, logit-opacities, wxyz quaternions."""
    means: Tensor       # [N,3]
    scales: Tensor      # [N,3] log
    quats: Tensor       # [N,4] wxyz, not normalised
    opacities: Tensor   # [N,1] logit
    colors: Tensor      # [N,3]

    def to(self, device):
        return SplatSet(*(t.to(device) for t in (self.means, self.scales, self.quats, self.opacities, self.colors)))

    @property
    def num(self) -> int:
        return self.means.shape[0]


def _bary2gs(p0, p1, area, normals, max_scale_ratio, g_scale_ratio=1.6):
    """rfstudio/model/geosplat.py:390-424"""
    means = (p0 + p1) / 2
    max_rots = p1 - means
    max_scales = max_rots.norm(dim=-1, keepdim=True).clamp(min=1e-10)
    min_scales = area / 4 / max_scales
    max_rots = max_rots / max_scales
    scales = torch.cat(((g_scale_ratio * max_scale_ratio * max_scales).log(),
                        (g_scale_ratio / max_scale_ratio * min_scales).log(),
                        torch.full_like(max_scales, -10.0)), dim=-1)
    min_rots = torch.cross(normals, max_rots, dim=-1)
    quats = rot2quat(torch.stack((max_rots, min_rots, normals), dim=-1))
    opac = torch.full_like(means[:, :1], 0.99).logit()
    return means, scales, quats, opac


def mesh_to_splats(vertices: Tensor, faces: Tensor, vnormals: Tensor) -> Tuple[SplatSet, Tensor]:
    """MGAdapter.make with its default ratios (rfstudio/model/geosplat.py:381-388,426-472):
    two rings of three Gaussians per face, colours = interpolated shading normals.
    Returns (splats, shading_normals[N,3])."""
    scale_ratio = (0.5, 1.3); l_scale_ratio = (1.0 / 3.0, 3.0); bias = (-1.0 / 24.0, 0.0)
    p0, p1, p2 = vertices[faces[:, 0]], vertices[faces[:, 1]], vertices[faces[:, 2]]
    vn0, vn1, vn2 = vnormals[faces[:, 0]], vnormals[faces[:, 1]], vnormals[faces[:, 2]]
    fn = torch.cross(p1 - p0, p2 - p0, dim=-1)
    area = fn.norm(dim=-1, keepdim=True).clamp(min=1e-10) / 2
    fn = safe_normalize(fn)
    parts, shading = [], []
    for u_coeff, a_coeff, s_ratio in zip([1 / 9 + bias[0], 2 / 9 + bias[1]],
                                         [1 / 4 * l_scale_ratio[0], 1 / 12 * l_scale_ratio[1]], scale_ratio):
        u0 = p0 * (1 - 2 * u_coeff) + (p1 + p2) * u_coeff
        u1 = p1 * (1 - 2 * u_coeff) + (p2 + p0) * u_coeff
        u2 = p2 * (1 - 2 * u_coeff) + (p0 + p1) * u_coeff
        n0 = vn0 * (1 - 2 * u_coeff) + (vn1 + vn2) * u_coeff
        n1 = vn1 * (1 - 2 * u_coeff) + (vn2 + vn0) * u_coeff
        n2 = vn2 * (1 - 2 * u_coeff) + (vn0 + vn1) * u_coeff
        a = area * a_coeff
        for (a0, a1, na, nb) in ((u0, u1, n0, n1), (u1, u2, n1, n2), (u2, u0, n2, n0)):
            parts.append(_bary2gs(a0, a1, a, fn, s_ratio))
            shading.append(safe_normalize((na + nb) / 2))
    means, scales, quats, opac = (torch.cat([p[i] for p in parts], 0) for i in range(4))
    normals = torch.cat(shading, 0)
    return SplatSet(means, scales, quats, opac, normals.clone()), normals


def random_quaternion(n: int, gen: torch.Generator) -> Tensor:
    """rfstudio/graphics/math.py:59-72"""
    u = torch.rand(n, generator=gen); v = torch.rand(n, generator=gen) * (2 * math.pi)
    w = torch.rand(n, generator=gen) * (2 * math.pi)
    return torch.stack([torch.sqrt(1 - u) * torch.sin(v), torch.sqrt(1 - u) * torch.cos(v),
                        torch.sqrt(u) * torch.sin(w), torch.sqrt(u) * torch.cos(w)], -1)


def random_splats(n: int, seed: int = 1, random_scale: float = 1.0) -> SplatSet:
    """Splats.random semantics (rfstudio/graphics/_splats.py:61-80): uniform positions in [-s,s]^3, isotropic
    log-scale = log(mean distance to the 3 nearest neighbours), random rotations, opacity logit(0.1);
    colours seeded U[0,1] instead of the constant 0.5 so that colour gradients are non-degenerate."""
    g = torch.Generator().manual_seed(seed)
    pos = (torch.rand(n, 3, generator=g) - 0.5) * (2 * random_scale)
    # 3-NN mean distance, chunked brute force (n <= a few 10k)
    d = torch.empty(n)
    for s in range(0, n, 2048):
        dist = torch.cdist(pos[s:s + 2048], pos)
        d[s:s + 2048] = dist.topk(4, largest=False).values[:, 1:].mean(-1)
    return SplatSet(pos, d.log()[:, None].repeat(1, 3), random_quaternion(n, g),
                    torch.logit(torch.full((n, 1), 0.1)), torch.rand(n, 3, generator=g))


@dataclass
class PBRScene:
    splats: SplatSet
    normals: Tensor     # [N,3]
    kd: Tensor          # [N,3]
    ks: Tensor          # [N,2]
    cubemap: Tensor     # [6,R,R,3]


def make_cubemap(res: int = 512, seed: int = 1) -> Tensor:
    """Seeded smooth HDR environment: sum of a few directional lobes + sky gradient, clamped >= 1e-2."""
    g = torch.Generator().manual_seed(seed)
    lin = torch.linspace(-1 + 1 / res, 1 - 1 / res, res)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    one = torch.ones_like(gx)
    tab = [(one, -gy, -gx), (-one, -gy, gx), (gx, one, gy), (gx, -one, -gy), (gx, -gy, one), (-gx, -gy, -one)]
    dirs = torch.stack([torch.stack(t, -1) for t in tab], 0)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    env = 0.15 + 0.35 * (dirs[..., 1:2] * 0.5 + 0.5) * torch.tensor([0.6, 0.8, 1.0])
    for _ in range(6):
        c = torch.randn(3, generator=g); c = c / c.norm()
        sharp = 4.0 + 60.0 * torch.rand(1, generator=g)
        col = 0.5 + 4.0 * torch.rand(3, generator=g)
        env = env + col * torch.exp(sharp * ((dirs * c).sum(-1, keepdim=True) - 1.0))
    return env.clamp_min(1e-2).float().contiguous()


def sphere_scene(level: int, seed: int = 1, radius: float = 0.8, cubemap_res: int = 512) -> PBRScene:
    """Bumpy icosphere -> MGAdapter splats + smooth procedural kd / seeded ks."""
    g = torch.Generator().manual_seed(seed)
    v, f = icosphere(level, 1.0)
    # low-frequency radial bumps
    bump = torch.zeros(v.shape[0])
    for _ in range(5):
        c = torch.randn(3, generator=g); c = c / c.norm()
        bump = bump + 0.04 * torch.cos(3.0 * (v @ c) + 6.28 * torch.rand(1, generator=g))
    v = v * (radius * (1.0 + bump))[:, None]
    vn = vertex_normals(v, f)
    splats, normals = mesh_to_splats(v, f, vn)
    p = splats.means
    kd = 0.5 + 0.45 * torch.stack([torch.sin(5 * p[:, 0] + 1.0), torch.sin(4 * p[:, 1] + 2.0),
                                   torch.sin(6 * p[:, 2] + 3.0)], -1)
    ks = torch.rand(p.shape[0], 2, generator=g)
    return PBRScene(splats, normals.contiguous(), kd.float().contiguous(), ks.float().contiguous(),
                    make_cubemap(cubemap_res, seed))


def blender_cameras(num: int = 8, width: int = 800, height: int = 800) -> List[Camera]:
    """Cameras as the Blender-style parsers produce them
    (rfstudio/data/dataparser/syn4relight_dataparser.py:42-75): camera_angle_x = 0.6911112."""
    focal = 0.5 * width / math.tan(0.5 * 0.6911112)
    return orbit_cameras(num, radius=4.0 * (2.0 / 3.0), pitch_degree=30.0, width=width, height=height, focal=focal)
