"""Seeded synthetic inputs of the shapes BASELINE.json names (no datasets exist on either box).

  * ``random_splats``      : config 0 -- semantics of Splats.random (rfstudio/graphics/_splats.py:61-80)
  * ``icosphere`` + ``sphere_scene`` : configs 1..3 -- GeoSplatting-like surface splats: a bumpy icosphere pushed
    through the product's own HIP MGAdapter (``geosplatting_amd.mesh``: 6 flat Gaussians per face);
    level 6 -> 491 520 Gaussians, level 7 -> 1 966 080 Gaussians
  * ``blender_cameras``    : 800x800, fx=fy=0.5*800/tan(0.5*0.6911112), eye distance 4*(2/3)
                             (rfstudio/data/dataparser/syn4relight_dataparser.py:42-75)
  * ``make_cubemap``       : seeded smooth HDR environment >= 1e-2 (rfstudio/trainer/geosplat_trainer.py:266)

Meshes, cameras and textures are host-side torch; the Gaussians of ``sphere_scene`` come from the HIP kernels (a GPU is
required -- the CPU restatement of the mesh -> Gaussians step is test infrastructure and lives in oracle/mesh_ref.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import torch
from torch import Tensor

from .cameras import Camera, orbit_cameras
from .splats import SplatSet


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


def random_splats(n: int, seed: int = 1, random_scale: float = 1.0) -> SplatSet:
    """Inputs shaped like Splats.random (rfstudio/graphics/_splats.py:61-80): uniform positions in [-s,s]^3, isotropic
    log-scale = log(mean distance to the 3 nearest neighbours), uniformly distributed rotations (normalised 4-D normal
    deviates), opacity logit(0.1); colours seeded U[0,1] instead of the constant 0.5 so that colour gradients are
    non-degenerate."""
    g = torch.Generator().manual_seed(seed)
    pos = (torch.rand(n, 3, generator=g) - 0.5) * (2 * random_scale)
    d = torch.empty(n)
    for s in range(0, n, 2048):                                    # chunked brute-force 3-NN (n <= a few 10k)
        dist = torch.cdist(pos[s:s + 2048], pos)
        d[s:s + 2048] = dist.topk(4, largest=False).values[:, 1:].mean(-1)
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    return SplatSet(pos, d.log()[:, None].repeat(1, 3), q, torch.logit(torch.full((n, 1), 0.1)),
                    torch.rand(n, 3, generator=g))


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


def bumpy_sphere(level: int, seed: int = 1, radius: float = 0.8) -> Tuple[Tensor, Tensor, torch.Generator]:
    """Icosphere with seeded low-frequency radial bumps -> (vertices, faces, the generator for further draws)."""
    g = torch.Generator().manual_seed(seed)
    v, f = icosphere(level, 1.0)
    bump = torch.zeros(v.shape[0])
    for _ in range(5):
        c = torch.randn(3, generator=g); c = c / c.norm()
        bump = bump + 0.04 * torch.cos(3.0 * (v @ c) + 6.28 * torch.rand(1, generator=g))
    return v * (radius * (1.0 + bump))[:, None], f, g


def _hip_mesh_to_splats(device) -> Callable:
    from . import _lib
    from .mesh import mesh_to_splats, vertex_normals
    if not torch.cuda.is_available():
        raise _lib.GeoSplatHipError("sphere_scene builds its Gaussians with the HIP MGAdapter kernels: a GPU is required "
                                    "(tests without one use oracle/mesh_ref.py through tests/util.py)")
    dev = torch.device(device if device is not None else "cuda:0")

    def build(v: Tensor, f: Tensor):
        vd, fd = v.to(dev), f.to(dev)
        with torch.no_grad():
            sp, normals = mesh_to_splats(vd, fd, vertex_normals(vd, fd))
        return sp.means.cpu(), sp.scales.cpu(), sp.quats.cpu(), sp.opacities.cpu(), normals.cpu()
    return build


def sphere_scene(level: int, seed: int = 1, radius: float = 0.8, cubemap_res: int = 512, device=None,
                 mesh_to_splats_fn: Optional[Callable] = None) -> PBRScene:
    """Bumpy icosphere -> surface splats + smooth procedural kd / seeded ks, returned on the host.
    The mesh -> Gaussians step runs on the product's HIP kernels (``device``, default cuda:0);
    ``mesh_to_splats_fn(vertices, faces) -> (means, log_scales, quats, logit_opacities, normals)`` replaces it
    (the tests pass the CPU restatement from oracle/mesh_ref.py so that CPU-only suites can build scenes)."""
    v, f, g = bumpy_sphere(level, seed, radius)
    build = mesh_to_splats_fn if mesh_to_splats_fn is not None else _hip_mesh_to_splats(device)
    means, scales, quats, opac, normals = build(v, f)
    splats = SplatSet(means, scales, quats, opac, normals.clone())
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
