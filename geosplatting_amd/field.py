"""Hash-grid field encoder on the GPU: `HashEncoding` of
rfstudio/model/components/encoding.py:87-241 with its `backend='torch'` semantics (the branch a ROCm user of the
reference runs -- tinycudann is CUDA-only), as used for kd / ks / z per Gaussian at rfstudio/model/geosplat.py:482-520,
644-672.  The encoding (gather / interpolate / scatter) is hand-written HIP (csrc/gs_hashgrid.hip); the 32-wide MLP
behind it is two or three plain GEMMs and goes to the BLAS library through torch.  No CPU path.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

import ctypes as C

from . import _lib


def level_scalings(num_levels: int = 16, min_res: int = 16, max_res: int = 1024) -> Tensor:
    """encoding.py:124-132: floor(min_res * growth^l), evaluated with the same torch expression as the reference so
    that the float32 values (and with them every cell index) are identical."""
    levels = torch.arange(num_levels)
    growth = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
    return torch.floor(min_res * growth ** levels)


class _HashGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, table: Tensor, scalings: Sequence[float], log2_T: int, table_grad_scale: float):
        _lib.require_cuda(x, table)
        L = len(scalings)
        if x.ndim != 2 or x.shape[1] != 3:
            raise _lib.GeoSplatHipError("hash_encode expects x [N,3]")
        if table.ndim != 2 or table.shape[0] != L * (1 << log2_T):
            raise _lib.GeoSplatHipError(f"hash table must be [{L} * 2^{log2_T}, F]")
        xd = x.detach().contiguous().float(); td = table.detach().contiguous().float()
        N, F = xd.shape[0], td.shape[1]
        out = torch.empty(N, L * F, device=xd.device)
        sc = (C.c_float * L)(*[float(s) for s in scalings])
        _lib.check(_lib.lib().gs_hashgrid_fwd(N, L, F, log2_T, sc, _lib.ptr(xd), _lib.ptr(td), _lib.ptr(out), _lib.stream()),
                   "gs_hashgrid_fwd")
        ctx.save_for_backward(xd, td)
        ctx.cfg = (tuple(float(s) for s in scalings), log2_T, float(table_grad_scale))
        return out

    @staticmethod
    def backward(ctx, v_out: Tensor):
        xd, td = ctx.saved_tensors
        scalings, log2_T, tgs = ctx.cfg
        L = len(scalings)
        N, F = xd.shape[0], td.shape[1]
        v_table = torch.empty_like(td)
        v_x = torch.empty_like(xd) if ctx.needs_input_grad[0] else None
        sc = (C.c_float * L)(*scalings)
        mode = os.environ.get("GEOSPLAT_HASHGRID_SLABS", "2")   # 2: fixed-point slabs over binned points, 1: float slabs, 0: fp32 atomics
        lib = _lib.lib()
        vo = v_out.contiguous().float()
        fixed_bytes = lib.gs_hashgrid_bwd_fixed_ws_bytes(N, L, F, log2_T) if mode == "2" else 0
        if fixed_bytes:
            # the binned path reserves a worst-case queue of N ints per (level, slab) pair (4.3 GB at N = 2 M, 2^18 rows):
            # fine on a 288 GB MI355X, but never let it crowd out the model -- beyond a quarter of the free memory fall back to the
            # float-slab kernel, whose workspace is 4*N*L*F bytes
            budget_bytes = torch.cuda.mem_get_info(xd.device)[0] // 4
            if fixed_bytes > budget_bytes:
                fixed_bytes = 0
        if fixed_bytes:
            ws = torch.empty(fixed_bytes, dtype=torch.uint8, device=xd.device)
            _lib.check(lib.gs_hashgrid_bwd_fixed(N, L, F, log2_T, sc, _lib.ptr(xd), _lib.ptr(td), _lib.ptr(vo),
                                                 _lib.f32(tgs), _lib.ptr(v_table), 0, _lib.ptr(v_x), _lib.ptr(ws),
                                                 C.c_size_t(fixed_bytes), _lib.stream()), "gs_hashgrid_bwd_fixed")
            return v_x, v_table, None, None, None
        nbytes = lib.gs_hashgrid_bwd_ws_bytes(N, L, F) if mode != "0" else 0
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xd.device) if nbytes else None
        _lib.check(lib.gs_hashgrid_bwd(N, L, F, log2_T, sc, _lib.ptr(xd), _lib.ptr(td), _lib.ptr(vo),
                                       _lib.f32(tgs), _lib.ptr(v_table), 0, _lib.ptr(v_x), _lib.ptr(ws),
                                       C.c_size_t(nbytes), _lib.stream()), "gs_hashgrid_bwd")
        return v_x, v_table, None, None, None


def hash_encode(x: Tensor, table: Tensor, scalings: Tensor, log2_hashmap_size: int,
                grad_scaling: Optional[float] = None) -> Tensor:
    """`HashEncoding.pytorch_fwd` wrapped in the grad-scaling trick of `HashEncoding.__call__` (encoding.py:231-240):
    features [N, L*F]; with grad_scaling the TABLE gradient is multiplied by it, the input gradient is unchanged."""
    return _HashGrid.apply(x, table, [float(s) for s in scalings.tolist()], int(log2_hashmap_size),
                           1.0 if grad_scaling is None else float(grad_scaling))


class _Linear(torch.autograd.Function):
    """y = x W^T (bias-free layer of rfstudio/nn/mlp.py:126-145).  Forward and dX are library GEMMs (N x 32 x 32: fine);
    dW = dY^T X is a 32 x 32 x N product that the library runs at 2-5 ms for N = 2 M -- it goes through gs_mlp_wgrad."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor) -> Tensor:
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(x, w)

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, w = ctx.saved_tensors
        gx = gy @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            lib = _lib.lib()
            gyc, xc = gy.contiguous().float(), x.detach().contiguous().float()
            N, O, I = xc.shape[0], w.shape[0], w.shape[1]
            gw = torch.empty(O, I, device=w.device)
            nbytes = lib.gs_mlp_wgrad_ws_bytes(_lib.i64(N))
            ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=w.device)
            _lib.check(lib.gs_mlp_wgrad(_lib.i64(N), O, I, _lib.ptr(gyc), _lib.ptr(xc), _lib.f32(1.0), _lib.ptr(gw), 0,
                                        _lib.ptr(ws), C.c_size_t(nbytes), _lib.stream()), "gs_mlp_wgrad")
        return gx, gw


def linear(x: Tensor, w: Tensor) -> Tensor:
    """torch.nn.functional.linear(x, w); 2-D x with both widths <= 32 (every layer of the field) takes the HIP weight
    gradient, wider layers are plain library GEMMs in both directions."""
    if x.dim() == 2 and w.shape[0] <= 32 and w.shape[1] <= 32:
        _lib.require_cuda(x, w)
        return _Linear.apply(x.float(), w)
    return torch.nn.functional.linear(x, w)


class HashEncoding:
    """Mirror of the reference's HashEncoding + MLP pair (same field names): `enc(x)` = mlp(hash features)."""

    def __init__(self, mlp_layers: Sequence[int], activation: str = "none", num_levels: int = 16, min_res: int = 16,
                 max_res: int = 1024, log2_hashmap_size: int = 19, features_per_level: int = 2,
                 hash_init_scale: float = 0.001, grad_scaling: Optional[float] = None, device="cuda", seed: int = 0):
        if features_per_level != 2:
            raise NotImplementedError("features_per_level = 2 (rfstudio/model/geosplat.py:485-518)")
        if activation not in ("none", "sigmoid"):
            raise NotImplementedError(activation)
        g = torch.Generator().manual_seed(seed)
        self.num_levels, self.log2_hashmap_size, self.grad_scaling, self.activation = num_levels, log2_hashmap_size, grad_scaling, activation
        self.scalings = level_scalings(num_levels, min_res, max_res)
        T = 2 ** log2_hashmap_size
        self.hash_table = ((torch.rand(T * num_levels, features_per_level, generator=g) * 2 - 1) * hash_init_scale).to(device).requires_grad_(True)
        dims = [num_levels * features_per_level] + list(mlp_layers[1:])
        self.weights: List[Tensor] = []
        for i, o in zip(dims[:-1], dims[1:]):
            w = torch.empty(o, i)
            torch.nn.init.kaiming_uniform_(w, nonlinearity="relu", generator=g)     # MLP initialization='kaiming-uniform', bias=False
            self.weights.append(w.to(device).requires_grad_(True))

    def parameters(self) -> List[Tensor]:
        return [self.hash_table] + self.weights

    def state_dict(self) -> dict:
        """Keys of the reference module's state_dict: the table is `encoder.params` (a ParameterModule,
        rfstudio/model/components/encoding.py:144-148), layer i of the bias-free MLP `mlp.nn_layers.{i}.weight` ([out, in],
        rfstudio/nn/mlp.py:52-62) -- what `export_model` stores under 'ks_enc' and stage 2 loads with `load_state_dict`."""
        sd = {"encoder.params": self.hash_table.detach().cpu().clone()}
        for i, w in enumerate(self.weights):
            sd[f"mlp.nn_layers.{i}.weight"] = w.detach().cpu().clone()
        return sd

    def load_state_dict(self, sd: dict) -> None:
        want = set(self.state_dict().keys())
        if set(sd.keys()) != want:
            raise KeyError(f"HashEncoding.load_state_dict: expected keys {sorted(want)}, got {sorted(sd.keys())}")
        with torch.no_grad():
            if tuple(sd["encoder.params"].shape) != tuple(self.hash_table.shape):
                raise ValueError("hash table shape mismatch")
            self.hash_table.copy_(sd["encoder.params"].to(self.hash_table.device))
            for i, w in enumerate(self.weights):
                src = sd[f"mlp.nn_layers.{i}.weight"]
                if tuple(src.shape) != tuple(w.shape):
                    raise ValueError(f"layer {i} shape mismatch")
                w.copy_(src.to(w.device))

    def __call__(self, x: Tensor) -> Tensor:
        f = hash_encode(x, self.hash_table, self.scalings, self.log2_hashmap_size, self.grad_scaling)
        for i, w in enumerate(self.weights):
            f = linear(f, w)
            if i < len(self.weights) - 1:
                f = torch.relu(f)
            elif self.activation == "sigmoid":
                f = f.sigmoid()
        return f


# ----------------------------------------------------------------------------- 'vertex' sampling helpers (host-side glue, plain torch)
def _unit_or_z(v: Tensor) -> Tensor:
    """rfstudio.graphics.math.safe_normalize (:119-128): unit vectors, (0, 0, 1) where the length is below 1e-6."""
    n = v.norm(dim=-1, keepdim=True)
    z = torch.tensor([0.0, 0.0, 1.0], device=v.device, dtype=v.dtype)
    return torch.where(n < 1e-6, z, v / n.clamp_min(1e-6))


def vertex_patches(vertices: Tensor, faces: Tensor) -> Tuple[Tensor, Tensor]:
    """GaussianField.get_patches (rfstudio/model/geosplat.py:520-557): per vertex the normal (normalised sum of the UNIT normals of
    its faces -- unlike compute_vertex_normals, which weights by area) and a third of the area of its face fan projected on that
    normal, `sum_f (n_f |A_f| . n_v) / 6` with |A_f| twice the face area, floored at 1e-10 / 6.  Differentiable (torch ops:
    V Gaussians for the first 50 steps of a run -- nothing here is on the hot path)."""
    F = faces.shape[0]
    p = vertices[faces]                                                    # [F,3,3]
    wn = torch.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], dim=-1)           # area-weighted face normals
    flat = faces.reshape(-1)
    unit = _unit_or_z(wn)[:, None, :].expand(F, 3, 3).reshape(-1, 3)
    normals = _unit_or_z(torch.zeros_like(vertices).index_add_(0, flat, unit))
    proj = (wn[:, None, :] * normals[faces]).sum(-1).reshape(-1, 1)         # [3F,1]
    areas = torch.zeros_like(vertices[:, :1]).index_add_(0, flat, proj)
    return normals, areas.clamp_min(1e-10) / 6


def rotation_between(a: Tensor, b: Tensor, eps: float = 1e-6, generator: Optional[torch.Generator] = None) -> Tensor:
    """get_rotation_from_relative_vectors (rfstudio/graphics/math.py:159-188): the rotation [..,3,3] that turns `a` into `b`
    (Rodrigues from v = a x b:  I + [v]x + [v]x^2 (1 - c) / (|v|^2 + eps)).  Where the two are opposite (c < -1 + eps) the
    reference perturbs `a` by uniform noise of amplitude 0.005 and starts over; so does this (noise from `generator`)."""
    a = a / a.norm(dim=-1, keepdim=True)
    b = b / b.norm(dim=-1, keepdim=True)
    c = (a * b).sum(-1)
    bad = c < -1 + eps
    if bool(bad.any()):
        noise = (torch.rand(torch.broadcast_shapes(a.shape, b.shape), device=b.device, generator=generator) - 0.5) * 0.01
        return rotation_between(a + torch.where(bad[..., None], noise, torch.zeros_like(noise)), b, eps, generator)
    v = torch.cross(a.expand(*c.shape, 3), b.expand(*c.shape, 3), dim=-1)
    z = torch.zeros_like(c)
    K = torch.stack((z, -v[..., 2], v[..., 1], v[..., 2], z, -v[..., 0], -v[..., 1], v[..., 0], z), -1).reshape(*c.shape, 3, 3)
    k = (1 - c) / ((v * v).sum(-1).sqrt() ** 2 + eps)
    return torch.eye(3, device=b.device, dtype=b.dtype) + K + (K @ K) * k[..., None, None]


def quaternions_from_rotations(R: Tensor) -> Tensor:
    """rot2quat (rfstudio/graphics/math.py:246-278): wxyz from [..,3,3], taking the best-conditioned of the four candidate rows
    (largest of 1 +- m00 +- m11 +- m22), denominators floored at 0.1."""
    m = R.reshape(-1, 3, 3)
    d0, d1, d2 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    four = torch.stack((1 + d0 + d1 + d2, 1 + d0 - d1 - d2, 1 - d0 + d1 - d2, 1 - d0 - d1 + d2), -1)
    mag = four.clamp_min(0).sqrt()
    pick = mag.argmax(-1)
    s_yz, s_zx, s_xy = m[:, 2, 1] - m[:, 1, 2], m[:, 0, 2] - m[:, 2, 0], m[:, 1, 0] - m[:, 0, 1]
    a_xy, a_zx, a_yz = m[:, 1, 0] + m[:, 0, 1], m[:, 0, 2] + m[:, 2, 0], m[:, 1, 2] + m[:, 2, 1]
    sq = mag * mag
    rows = torch.stack((torch.stack((sq[:, 0], s_yz, s_zx, s_xy), -1), torch.stack((s_yz, sq[:, 1], a_xy, a_zx), -1),
                        torch.stack((s_zx, a_xy, sq[:, 2], a_yz), -1), torch.stack((s_xy, a_zx, a_yz, sq[:, 3]), -1)), 1)   # [n,4,4]
    q = rows[torch.arange(m.shape[0], device=m.device), pick] / (2 * mag.gather(1, pick[:, None]).clamp_min(0.1))
    return q.reshape(*R.shape[:-2], 4)


class GaussianField:
    """Mirror of rfstudio's GaussianField (rfstudio/model/geosplat.py:482-520) with its three default encoders and of
    `get_gaussians_from_face` (:620-672, the MGAdapter branch used by GeoSplatter): mesh -> Gaussians, with kd / ks
    from the hash-grid field at the Gaussian centres and a learned offset of the centres against the face normal.
    Every stage is a HIP op of this package (vertex normals, MGAdapter, hash encoding) or a library GEMM."""

    def __init__(self, device="cuda", log2_hashmap_size: int = 18, max_res: int = 4096, seed: int = 0):
        common = dict(max_res=max_res, log2_hashmap_size=log2_hashmap_size, grad_scaling=16.0, device=device)
        self.kd_enc = HashEncoding([-1, 32, 32, 3], activation="sigmoid", seed=seed, **common)
        self.ks_enc = HashEncoding([-1, 32, 2], activation="none", seed=seed + 1, **common)
        self.z_enc = HashEncoding([-1, 32, 1], activation="none", seed=seed + 2, **common)

    def parameters(self) -> List[Tensor]:
        return self.kd_enc.parameters() + self.ks_enc.parameters() + self.z_enc.parameters()

    def get_gaussians_from_vertex(self, vertices: Tensor, faces: Tensor, kd_perturb_std: float = 0.0, ks_perturb_std: float = 0.0, *,
                                  scale: float, initial_guess: Tensor, generator: Optional[torch.Generator] = None):
        """GaussianField.get_gaussians_from_vertex (rfstudio/model/geosplat.py:559-620), the sampling of the first
        `vertex_sample_warmup` steps: ONE flat Gaussian per mesh vertex, facing the vertex normal, its two in-plane log-scales
        0.5 log(patch area / 2.5) (the third log 1e-10), pushed under the surface by sigmoid(z field) times that scale.  Returns
        (SplatSet, RenderableAttrs)."""
        from .shading import RenderableAttrs
        from .splats import SplatSet
        normals, areas = vertex_patches(vertices, faces)
        half_log = (areas * (1 / 2.5)).log() * 0.5                              # [V,1]
        x = (vertices / scale).clamp(-1, 1)
        kd_jitter = ks_jitter = None
        if kd_perturb_std > 0:
            kd_jitter = self.kd_enc((x + torch.randn(x.shape, device=x.device, generator=generator) * kd_perturb_std).clamp(-1, 1))
        if ks_perturb_std > 0:
            ks_jitter = (self.ks_enc((x + torch.randn(x.shape, device=x.device, generator=generator) * ks_perturb_std).clamp(-1, 1)) + initial_guess).sigmoid()
        attrs = RenderableAttrs(kd=self.kd_enc(x), ks=(self.ks_enc(x) + initial_guess).sigmoid(), normals=normals,
                                kd_jitter=kd_jitter, ks_jitter=ks_jitter)
        depth = half_log.detach().exp() * self.z_enc(x.detach()).sigmoid()        # [V,1]
        z_axis = torch.tensor([0.0, 0.0, 1.0], device=vertices.device, dtype=vertices.dtype)
        quats = quaternions_from_rotations(rotation_between(z_axis, normals.detach(), generator=generator))
        scales = torch.cat((half_log, half_log, torch.full_like(half_log, 1e-10).log()), -1)
        V = vertices.shape[0]
        opac = torch.logit(0.99 * torch.ones(V, 1, device=vertices.device))
        return SplatSet(vertices - normals * depth, scales, quats, opac, torch.empty_like(normals)), attrs

    def get_gaussians_from_face(self, vertices: Tensor, faces: Tensor, kd_perturb_std: float = 0.0,
                                ks_perturb_std: float = 0.0, *, scale: float, initial_guess: Tensor,
                                generator: Optional[torch.Generator] = None):
        """Returns (SplatSet with the shifted means, RenderableAttrs, offsets[6F,3])."""
        from .mesh import mesh_to_splats, vertex_normals
        from .shading import RenderableAttrs
        from .splats import SplatSet
        splats, shading_normals = mesh_to_splats(vertices, faces, vertex_normals(vertices, faces))
        with torch.no_grad():                                  # MGAdapter.make: offsets = n.detach() * sqrt(area.detach())
            p = vertices[faces]
            fn = torch.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], dim=-1)
            ln = fn.norm(dim=-1, keepdim=True)
            area = ln.clamp(min=1e-10) / 2
            n = torch.where(ln < 1e-6, torch.tensor([0.0, 0.0, 1.0], device=fn.device), fn / ln.clamp_min(1e-6))
            offsets = (n * area.sqrt()).repeat(6, 1)
        means = (splats.means / scale).clamp(-1, 1)
        offsets = offsets * self.z_enc(means.detach()).sigmoid()
        shifted = splats.means - offsets
        kd_jitter = ks_jitter = None
        if kd_perturb_std > 0:
            kd_jitter = self.kd_enc((means + torch.randn(means.shape, device=means.device, generator=generator) * kd_perturb_std).clamp(-1, 1))
        if ks_perturb_std > 0:
            ks_jitter = (self.ks_enc((means + torch.randn(means.shape, device=means.device, generator=generator) * ks_perturb_std).clamp(-1, 1)) + initial_guess).sigmoid()
        attrs = RenderableAttrs(kd=self.kd_enc(means), ks=(self.ks_enc(means) + initial_guess).sigmoid(), normals=shading_normals,
                                kd_jitter=kd_jitter, ks_jitter=ks_jitter)
        return SplatSet(shifted, splats.scales, splats.quats, splats.opacities, splats.colors), attrs, offsets
