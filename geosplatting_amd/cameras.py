"""Camera conventions of the reference (host-side glue, plain torch).

Mirrors the part of ``rfstudio.graphics.Cameras`` that the hot path consumes:
  * ``c2w`` is [3,4], OpenGL convention (x right, y up, -z forward)      rfstudio/graphics/_cameras.py:120-135
  * ``view_matrix``   : flip y/z to the OpenCV convention the rasterizer uses, analytic inverse
                                                                          rfstudio/graphics/_cameras.py:299-314
  * ``intrinsic_matrix`` : [[fx,0,cx],[0,fy,cy],[0,0,1]]                   rfstudio/graphics/_cameras.py:289-297
  * ``from_lookat`` / ``from_orbit`` constructors                         rfstudio/graphics/_cameras.py:68-167
Pinned against the importable reference by tests/golden/ref_cameras.npz.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch
from torch import Tensor


@dataclass
class Camera:
    """One pinhole camera (the reference asserts ``cameras.shape == (1,)`` on the path)."""
    c2w: Tensor          # [3,4]
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int

    def to(self, device) -> "Camera":
        return Camera(self.c2w.to(device), self.fx, self.fy, self.cx, self.cy, self.width, self.height)

    @property
    def view_matrix(self) -> Tensor:
        return view_matrix(self.c2w)

    @property
    def intrinsic_matrix(self) -> Tensor:
        return intrinsic_matrix(self.fx, self.fy, self.cx, self.cy, device=self.c2w.device)

    @property
    def position(self) -> Tensor:
        return self.c2w[:, 3]


def view_matrix(c2w: Tensor) -> Tensor:
    """world->camera 4x4 in OpenCV convention (rfstudio/graphics/_cameras.py:299-314)."""
    R = c2w[..., :3, :3]
    T = c2w[..., :3, 3:4]
    R = R * torch.tensor([1.0, -1.0, -1.0], device=R.device, dtype=R.dtype)
    R_inv = R.transpose(-1, -2)
    T_inv = R_inv @ -T
    out = c2w.new_zeros(c2w.shape[:-2] + (4, 4))
    out[..., 3, 3] = 1.0
    out[..., :3, :3] = R_inv
    out[..., :3, 3:4] = T_inv
    return out


def intrinsic_matrix(fx, fy, cx, cy, device=None, dtype=torch.float32) -> Tensor:
    K = torch.zeros(3, 3, device=device, dtype=dtype)
    K[0, 0] = fx; K[1, 1] = fy; K[0, 2] = cx; K[1, 2] = cy; K[2, 2] = 1.0
    return K


def lookat_c2w(eye: Tensor, target: Tensor, up: Tensor) -> Tensor:
    """c2w [...,3,4] as built by Cameras.from_lookat (rfstudio/graphics/_cameras.py:120-127)."""
    forward = target - eye
    right = torch.cross(forward, up, dim=-1)
    up2 = torch.cross(right, forward, dim=-1)
    R = torch.stack((right, up2, -forward), dim=-1)
    R = R / R.norm(dim=-2, keepdim=True)
    return torch.cat((R, eye[..., None]), dim=-1)


def _spherical_positions(yaw: Tensor, pitch: float, up: Tensor, radius: float) -> Tensor:
    """Positions on a circle of elevation `pitch` around the `up` axis (orbit sampling)."""
    up = up / up.norm()
    # two tangents orthogonal to up
    helper = torch.tensor([1.0, 0.0, 0.0]) if abs(float(up[0])) < 0.9 else torch.tensor([0.0, 1.0, 0.0])
    t0 = torch.cross(up, helper, dim=-1); t0 = t0 / t0.norm()
    t1 = torch.cross(up, t0, dim=-1)
    cp, sp = math.cos(pitch), math.sin(pitch)
    return radius * (cp * (torch.cos(yaw)[:, None] * t0 + torch.sin(yaw)[:, None] * t1) + sp * up)


def orbit_cameras(num: int, radius: float, pitch_degree: float, width: int, height: int,
                  hfov_degree: Optional[float] = None, focal: Optional[float] = None,
                  center: Sequence[float] = (0.0, 0.0, 0.0), up: Sequence[float] = (0.0, 1.0, 0.0)):
    """`num` cameras on an orbit looking at `center` (semantics of Cameras.from_orbit,
    rfstudio/graphics/_cameras.py:129-167; cx=W/2, cy=H/2, fx=fy)."""
    upv = torch.tensor(up, dtype=torch.float32)
    c = torch.tensor(center, dtype=torch.float32)
    yaw = (2 * math.pi / num) * torch.arange(num, dtype=torch.float32)
    eyes = _spherical_positions(yaw, pitch_degree * math.pi / 180, upv, radius) + c
    c2w = lookat_c2w(eyes, c.expand(num, 3), upv.expand(num, 3))
    cx, cy = width * 0.5, height * 0.5
    if focal is None:
        focal = cx / math.tan(hfov_degree * (0.5 * math.pi / 180))
    return [Camera(c2w[i].contiguous(), float(focal), float(focal), cx, cy, width, height) for i in range(num)]
