"""FlexiCubes extraction on the GPU: the differentiable SDF-grid -> triangle-mesh step that
runs in front of MGAdapter every stage-1 iteration (rfstudio/model/geosplat.py:751-769 `get_geometry`).

`FlexiCubes` mirrors the reference dataclass (rfstudio/graphics/_mesh/_flexicubes.py:369-457): same fields,
`from_resolution`, `replace`, `dual_marching_cubes(sdf_eps=None, weight_scale=0.99) -> ((vertices, faces), L_dev)` and
`compute_entropy()`; vertex / face / L_dev numbering is the reference's.  `grad_func` (the non-differentiable QEF
variant) raises NotImplementedError exactly as the reference does.  Gradients reach `vertices`, `sdf_values`,
`alpha`, `beta`, `gamma` through csrc/gs_flexicubes.hip.  No CPU path.

The grid must be the regular one `from_resolution` builds (the reference's ambiguity pass assumes it too, :468); the
kernels use the closed-form index layout, `indices` is kept for interface compatibility only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, replace as _dc_replace
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib


def _grid_indices(res: Tuple[int, int, int], device) -> Tensor:
    R0, R1, R2 = res
    n = torch.arange(R0 * R1 * R2, device=device)
    base = torch.stack((n % R0, (n // R0) % R1, n // (R0 * R1)), -1)[:, None, :]
    k = torch.arange(8, device=device)
    cc = base + torch.stack((k & 1, (k >> 1) & 1, (k >> 2) & 1), -1)[None]
    return (cc[..., 2] * (1 + R1) + cc[..., 1]) * (1 + R0) + cc[..., 0]


class _Extract(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, sdf, alpha, beta, gamma, res, weight_scale, sdf_eps):
        R0, R1, R2 = res
        lib = _lib.lib()
        dev = vertices.device
        Vg, Cn = (R0 + 1) * (R1 + 1) * (R2 + 1), R0 * R1 * R2
        f32c = lambda t: None if t is None else t.detach().contiguous().float()
        v, s, a, b, g = f32c(vertices), f32c(sdf).reshape(-1), f32c(alpha), f32c(beta), f32c(gamma)
        if v.shape != (Vg, 3) or s.numel() != Vg:
            raise _lib.GeoSplatHipError(f"vertices / sdf_values do not match the {R0}x{R1}x{R2} grid")
        for t, w, name in ((a, 8, "alpha"), (b, 12, "beta"), (g, 1, "gamma")):
            if t is not None and t.numel() != Cn * w:
                raise _lib.GeoSplatHipError(f"{name} must be [{Cn}, {w}]")
        ws_bytes = lib.gs_flexicubes_ws_bytes(R0, R1, R2)
        if ws_bytes == 0:
            raise _lib.GeoSplatHipError("resolution out of range")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        counts = torch.empty(8, dtype=torch.int64, device=dev)
        _lib.check(lib.gs_flexicubes_count(R0, R1, R2, _lib.ptr(s), _lib.ptr(ws), C.c_size_t(ws_bytes), _lib.ptr(counts),
                                           _lib.stream()), "gs_flexicubes_count")
        N, Q, K, E, nq, _, _, _ = counts.tolist()                    # the reference syncs here too (:605, :651)
        if N <= 0:                                                        # the reference asserts here (:606)
            raise AssertionError("no surface cube: the sdf does not change sign anywhere on the grid")
        out_v = torch.empty(Q + nq, 3, device=dev); faces = torch.empty(4 * nq, 3, dtype=torch.int64, device=dev)
        L_dev = torch.empty(K, device=dev)
        eps = -1.0 if sdf_eps is None else float(sdf_eps)
        _lib.check(lib.gs_flexicubes_fwd(R0, R1, R2, _lib.ptr(v), _lib.ptr(s), _lib.ptr(a), _lib.ptr(b), _lib.ptr(g),
                                         _lib.f32(weight_scale), _lib.f32(eps), _lib.ptr(ws), C.c_size_t(ws_bytes), _lib.i64(Q),
                                         _lib.i64(nq), _lib.i64(K), _lib.ptr(out_v), _lib.ptr(faces), _lib.ptr(L_dev),
                                         _lib.stream()), "gs_flexicubes_fwd")
        ctx.save_for_backward(v, s, a, b, g, ws, out_v)
        ctx.meta = (res, weight_scale, eps, Q, nq, K, ws_bytes, sdf.shape,
                    None if alpha is None else alpha.shape, None if beta is None else beta.shape,
                    None if gamma is None else gamma.shape)
        ctx.mark_non_differentiable(faces)
        return out_v, faces, L_dev

    @staticmethod
    def backward(ctx, v_out, _v_faces, v_L):
        v, s, a, b, g, ws, out_v = ctx.saved_tensors
        (R0, R1, R2), weight_scale, eps, Q, nq, K, ws_bytes, s_shape, a_shape, b_shape, g_shape = ctx.meta
        dev = v.device
        v_out = torch.zeros_like(out_v) if v_out is None else v_out.contiguous().float()
        v_L = None if v_L is None else v_L.contiguous().float()
        gv = torch.empty_like(v); gs = torch.empty_like(s)
        ga = None if a is None else torch.empty_like(a)
        gb = None if b is None else torch.empty_like(b)
        gg = None if g is None else torch.empty_like(g)
        scratch = torch.empty(Q, 3, device=dev)
        _lib.check(_lib.lib().gs_flexicubes_bwd(R0, R1, R2, _lib.ptr(v), _lib.ptr(s), _lib.ptr(a), _lib.ptr(b), _lib.ptr(g),
                                                _lib.f32(weight_scale), _lib.f32(eps), _lib.ptr(ws), C.c_size_t(ws_bytes),
                                                _lib.i64(Q), _lib.i64(nq), _lib.i64(K), _lib.ptr(out_v), _lib.ptr(v_out),
                                                _lib.ptr(v_L), _lib.ptr(scratch), _lib.ptr(gv), _lib.ptr(gs), _lib.ptr(ga),
                                                _lib.ptr(gb), _lib.ptr(gg), _lib.stream()), "gs_flexicubes_bwd")
        rs = lambda t, shp: None if t is None else t.reshape(shp)
        return gv, gs.reshape(s_shape), rs(ga, a_shape), rs(gb, b_shape), rs(gg, g_shape), None, None, None


class _Entropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, res):
        R0, R1, R2 = res
        lib = _lib.lib()
        s = sdf.detach().contiguous().float().reshape(-1)
        if s.numel() != (R0 + 1) * (R1 + 1) * (R2 + 1):
            raise _lib.GeoSplatHipError("sdf_values do not match the grid")
        ws_bytes = lib.gs_flexicubes_ws_bytes(R0, R1, R2)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=s.device)
        counts = torch.empty(8, dtype=torch.int64, device=s.device)
        out = torch.empty(1, device=s.device)
        _lib.check(lib.gs_flexicubes_count(R0, R1, R2, _lib.ptr(s), _lib.ptr(ws), C.c_size_t(ws_bytes), _lib.ptr(counts),
                                           _lib.stream()), "gs_flexicubes_count")
        _lib.check(lib.gs_flexicubes_entropy_fwd(R0, R1, R2, _lib.ptr(s), _lib.ptr(ws), C.c_size_t(ws_bytes), _lib.ptr(out),
                                                 _lib.stream()), "gs_flexicubes_entropy_fwd")
        ctx.save_for_backward(s, ws)
        ctx.meta = (res, ws_bytes, sdf.shape)
        return out.reshape(())

    @staticmethod
    def backward(ctx, v_out):
        s, ws = ctx.saved_tensors
        (R0, R1, R2), ws_bytes, shape = ctx.meta
        gs = torch.empty_like(s)
        _lib.check(_lib.lib().gs_flexicubes_entropy_bwd(R0, R1, R2, _lib.ptr(s), _lib.ptr(ws), C.c_size_t(ws_bytes),
                                                        _lib.ptr(v_out.reshape(1).contiguous().float()), _lib.ptr(gs), 0,
                                                        _lib.stream()), "gs_flexicubes_entropy_bwd")
        return gs.reshape(shape), None


@dataclass
class FlexiCubes:
    """Field-for-field the reference dataclass (`_flexicubes.py:369-396`)."""
    vertices: Tensor                       # [V,3] grid vertex positions (deformable)
    sdf_values: Tensor                     # [V,1]
    indices: Tensor                        # [C,8] corner ids of every cube (regular layout)
    resolution: Tensor                     # [3] int64
    alpha: Optional[Tensor] = None         # [C,8]  raw corner weights
    beta: Optional[Tensor] = None          # [C,12] raw edge weights
    gamma: Optional[Tensor] = None         # [C,1]  raw quad-split weights

    @classmethod
    def from_resolution(cls, *resolution: int, device=None, random_sdf: bool = True, scale: float = 1.0) -> "FlexiCubes":
        """`from_resolution` (:397-457): unit grid centred at the origin scaled by `scale`; sdf = U(-0.1, 0.9) or zeros."""
        assert len(resolution) in (1, 3)
        res = (resolution[0],) * 3 if len(resolution) == 1 else tuple(resolution)
        dev = torch.device("cuda") if device is None else torch.device(device)
        ax = [torch.arange(r + 1, device=dev) for r in res]
        coords = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3).float()
        rt = torch.tensor(res, dtype=torch.long, device=dev)
        verts = coords / rt
        sdf = (torch.rand_like(verts[..., 0:1]) - 0.1) if random_sdf else torch.zeros_like(verts[..., 0:1])
        out = cls(vertices=(2 * verts - 1) * scale, sdf_values=sdf, indices=_grid_indices(res, dev), resolution=rt)
        out.__dict__["_res_cached"] = tuple(int(r) for r in res)
        return out

    def replace(self, **kw) -> "FlexiCubes":
        out = _dc_replace(self, **kw)
        if "resolution" not in kw and "_res_cached" in self.__dict__:
            out.__dict__["_res_cached"] = self.__dict__["_res_cached"]
        return out

    @property
    def _res(self) -> Tuple[int, int, int]:
        if "_res_cached" not in self.__dict__:                              # one device read per grid, not per call
            r = self.resolution.tolist()
            self.__dict__["_res_cached"] = (int(r[0]), int(r[1]), int(r[2]))
        return self.__dict__["_res_cached"]

    def dual_marching_cubes(self, *, grad_func=None, sdf_eps: Optional[float] = None, weight_scale: float = 0.99
                            ) -> Tuple[Tuple[Tensor, Tensor], Tensor]:
        """`dual_marching_cubes` (:559-713) -> ((vertices [Q+quads,3], faces [4 quads,3] int64), L_dev [K])."""
        if grad_func is not None:
            raise NotImplementedError                                      # :644-645
        _lib.require_cuda(self.vertices, self.sdf_values)
        g = None if self.gamma is None else self.gamma
        v, f, L = _Extract.apply(self.vertices, self.sdf_values, self.alpha, self.beta, g, self._res, float(weight_scale),
                                 sdf_eps)
        return (v, f), L

    def compute_entropy(self) -> Tensor:
        """`compute_entropy` (:715-725)."""
        _lib.require_cuda(self.sdf_values)
        return _Entropy.apply(self.sdf_values, self._res)


def get_geometry(grid: FlexiCubes, deform_params: Tensor, sdf_params: Tensor, weight_params: Tensor, *, scale: float,
                 resolution: int, sdf_weight: float) -> Tuple[Tuple[Tensor, Tensor], Tensor]:
    """`GeoSplatter.get_geometry` (rfstudio/model/geosplat.py:751-769): deformed grid -> mesh + regulariser
    `L_dev.mean() * 0.5 + |alpha, beta|.mean() * 0.1 + entropy * sdf_weight`.  weight_params is [C, 21] = alpha | beta | gamma."""
    vertices = grid.vertices + deform_params.tanh() * (0.5 * scale / resolution)
    fc = grid.replace(vertices=vertices, sdf_values=sdf_params, alpha=weight_params[:, :8], beta=weight_params[:, 8:20],
                      gamma=weight_params[:, 20:])
    mesh, L_dev = fc.dual_marching_cubes()
    reg = torch.add(L_dev.mean() * 0.5 + weight_params[:, :20].abs().mean() * 0.1, fc.compute_entropy() * sdf_weight)
    return mesh, reg
