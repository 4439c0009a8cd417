"""MGAdapter on the GPU: mesh faces -> flat Gaussians, the step that feeds the render
path every iteration (rfstudio/model/geosplat.py:378-472, called from :843-868).

`mesh_to_splats(vertices, faces, vnormals)` has the semantics of `MGAdapter.make` with its default ratios and
returns `(SplatSet, shading_normals)`; gradients flow to `vertices` and `vnormals` through one HIP kernel each
way (csrc/gs_mesh.hip: forward-mode dual numbers inside the backward kernel, fp32 atomics into the vertices).
No CPU path: the torch restatement used for parity is test infrastructure (oracle/mesh_ref.py).
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
from torch import Tensor

from . import _lib
from .splats import SplatSet


class _MGAdapter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices: Tensor, faces: Tensor, vnormals: Tensor):
        _lib.require_cuda(vertices, faces, vnormals)
        if faces.dtype != torch.int64 or faces.ndim != 2 or faces.shape[1] != 3:
            raise _lib.GeoSplatHipError("faces must be int64 [F,3]")
        if vertices.shape != vnormals.shape or vertices.ndim != 2 or vertices.shape[1] != 3:
            raise _lib.GeoSplatHipError("vertices and vnormals must both be [V,3]")
        v = vertices.detach().contiguous().float(); n = vnormals.detach().contiguous().float(); f = faces.contiguous()
        F = f.shape[0]
        means = torch.empty(6 * F, 3, device=v.device); scales = torch.empty_like(means)
        quats = torch.empty(6 * F, 4, device=v.device); normals = torch.empty_like(means)
        _lib.check(_lib.lib().gs_mgadapter_fwd(F, _lib.ptr(v), _lib.ptr(f), _lib.ptr(n), _lib.ptr(means),
                                               _lib.ptr(scales), _lib.ptr(quats), _lib.ptr(normals), _lib.stream()),
                   "gs_mgadapter_fwd")
        ctx.save_for_backward(v, f, n)
        return means, scales, quats, normals

    @staticmethod
    def backward(ctx, v_means, v_scales, v_quats, v_normals):
        v, f, n = ctx.saved_tensors
        F, V = f.shape[0], v.shape[0]
        z = lambda g, w: (torch.zeros(6 * F, w, device=v.device) if g is None else g.contiguous().float())
        v_means, v_scales, v_quats = z(v_means, 3), z(v_scales, 3), z(v_quats, 4)
        v_normals = None if v_normals is None else v_normals.contiguous().float()
        gv = torch.empty_like(v); gn = torch.empty_like(n)
        _lib.check(_lib.lib().gs_mgadapter_bwd(F, V, _lib.ptr(v), _lib.ptr(f), _lib.ptr(n), _lib.ptr(v_means),
                                               _lib.ptr(v_scales), _lib.ptr(v_quats), _lib.ptr(v_normals),
                                               _lib.ptr(gv), _lib.ptr(gn), _lib.stream()), "gs_mgadapter_bwd")
        return gv, None, gn


class _VertexNormals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices: Tensor, faces: Tensor):
        _lib.require_cuda(vertices, faces)
        if faces.dtype != torch.int64 or faces.ndim != 2 or faces.shape[1] != 3:
            raise _lib.GeoSplatHipError("faces must be int64 [F,3]")
        v = vertices.detach().contiguous().float(); f = faces.contiguous()
        raw = torch.empty_like(v); vn = torch.empty_like(v)
        _lib.check(_lib.lib().gs_vertex_normals_fwd(f.shape[0], v.shape[0], _lib.ptr(v), _lib.ptr(f), _lib.ptr(raw),
                                                    _lib.ptr(vn), _lib.stream()), "gs_vertex_normals_fwd")
        ctx.save_for_backward(v, f, raw)
        return vn

    @staticmethod
    def backward(ctx, v_vn):
        v, f, raw = ctx.saved_tensors
        gv = torch.empty_like(v); scratch = torch.empty_like(v)
        _lib.check(_lib.lib().gs_vertex_normals_bwd(f.shape[0], v.shape[0], _lib.ptr(v), _lib.ptr(f), _lib.ptr(raw),
                                                    _lib.ptr(v_vn.contiguous().float()), _lib.ptr(scratch),
                                                    _lib.ptr(gv), 0, _lib.stream()), "gs_vertex_normals_bwd")
        return gv, None


def vertex_normals(vertices: Tensor, faces: Tensor) -> Tensor:
    """TriangleMesh.compute_vertex_normals(fix=True) (rfstudio/graphics/_mesh/_triangle_mesh.py:588-613)."""
    return _VertexNormals.apply(vertices, faces)


def mesh_to_splats(vertices: Tensor, faces: Tensor, vnormals: Tensor) -> Tuple[SplatSet, Tensor]:
    """MGAdapter.make (rfstudio/model/geosplat.py:426-472): returns (splats with colours = shading normals,
    shading_normals[6F,3]); row = part * F + face."""
    means, scales, quats, normals = _MGAdapter.apply(vertices, faces, vnormals)
    opac = torch.full((means.shape[0], 1), math.log(0.99 / 0.01), device=means.device)
    return SplatSet(means, scales, quats, opac, normals.clone()), normals
