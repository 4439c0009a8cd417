"""Shared helpers for the parity tests (scene builders, tolerances)."""
import math

import numpy as np
import torch

import geosplatting_amd.synthetic as syn
from oracle import mesh_ref
from geosplatting_amd.cameras import orbit_cameras


def rel_err(a, b):
    """max |a-b| / max |b|  (the '1e-4 relative fp32' of BASELINE.json is read as max-norm relative)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def elem_frac(a, b, rtol=1e-4, atol_rel=1e-6):
    """element-wise figure: fraction of elements with |a-b| > rtol*|b| + atol_rel*max|b|"""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float((np.abs(a - b) > rtol * np.abs(b) + atol_rel * np.abs(b).max()).mean())


def activated(splats):
    """numpy (means, quats, scales_exp, opacities_sigmoid) like GSplatter.render_rgba does (rfstudio/model/gsplat.py:336-339)"""
    return (splats.means.numpy(), splats.quats.numpy(), splats.scales.exp().numpy(),
            torch.sigmoid(splats.opacities).squeeze(-1).numpy())


def sphere_case(level, res, view=1, seed=1, cubemap_res=64):
    # Gaussians from the CPU restatement (oracle/mesh_ref.py): the same scene with and without a GPU; the product's own
    # HIP-built scene is compared with it in tests/test_gpu_mesh.py::test_sphere_scene_hip_equals_restatement
    sc = syn.sphere_scene(level, seed=seed, cubemap_res=cubemap_res, mesh_to_splats_fn=mesh_ref.scene_builder)
    focal = 0.5 * res / math.tan(0.5 * 0.6911112)
    cam = orbit_cameras(8, 4.0 * (2.0 / 3.0), 30.0, res, res, focal=focal)[view]
    return sc, cam


def random_case(n, res, view=0, seed=1):
    sp = syn.random_splats(n, seed=seed)
    cam = orbit_cameras(4, 3.0, 30.0, res, res, hfov_degree=40.0)[view]
    return sp, cam
