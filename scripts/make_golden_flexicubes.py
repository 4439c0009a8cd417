#!/usr/bin/env python3
"""Generate tests/golden/ref_flexicubes.npz from the reference's OWN FlexiCubes extraction (build container only):

    cd /tmp && PYTHONPATH=/tmp/stubs:/root/reference python /root/repo/scripts/make_golden_flexicubes.py

What it pins: FlexiCubes.from_resolution / dual_marching_cubes / compute_entropy
(rfstudio/graphics/_mesh/_flexicubes.py:397-457, 559-713, 715-725) executed as they are, on small grids that reach every
ambiguity branch (random signs), a non-cubic grid, a smooth blob and the no-weights call, plus the gradients torch
autograd gives the reference for a seeded cotangent.  It also asserts that the tables oracle/flexicubes_ref.py DERIVES
equal the reference's 256-case tables entry for entry (the tables themselves are not written anywhere).
Only input/output vectors are written.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
from oracle import flexicubes_ref as O                                    # noqa: E402

for name in ["open3d", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
             "cv2", "pyexr", "trimesh", "nvdiffrast", "nvdiffrast.torch", "kornia", "kornia.filters", "gsplat",
             "torchmetrics", "torchmetrics.functional", "torchmetrics.functional.image", "ffmpegcv", "nerfacc", "tyro",
             "skimage", "skimage.measure", "rfviser", "viser", "appdirs", "huggingface_hub", "pytorch3d",
             "pytorch3d.loss", "pytorch3d.structures", "rfstudio.graphics._mesh._optix", "rfstudio.graphics._mesh._splitsum",
             "tinycudann", "plotext", "imageio", "lpips", "matplotlib", "matplotlib.pyplot", "viser.transforms",
             "rfviser.transforms", "torchmetrics.image", "torchmetrics.image.lpip"]:
    sys.modules.setdefault(name, MagicMock())

from rfstudio.graphics._mesh import _flexicubes as R                      # noqa: E402

cpu = torch.device("cpu")
ref_dmc = R._get_dmc_table(cpu).numpy(); ref_nvd = R._get_num_vd_table(cpu).numpy(); ref_chk = R._get_check_table(cpu).numpy()
dmc, nvd, chk = O.tables()
for c in range(256):
    rows = [[int(e) for e in p if e >= 0] for p in ref_dmc[c]]
    assert [p for p in rows if p] == dmc[c], f"derived patch table differs at case {c}"
assert (nvd == ref_nvd).all() and (chk == ref_chk).all()
print("derived tables == reference tables")

CASES = [("rand4", (4, 4, 4), 0, True, False), ("rand6", (6, 6, 6), 1, True, False), ("rand567", (5, 6, 7), 2, True, False),
         ("blob10", (10, 10, 10), 3, True, True), ("blob_plain", (8, 9, 7), 4, False, True)]
out = {}
for tag, res, seed, weights, blob in CASES:
    torch.manual_seed(seed)
    fc = R.FlexiCubes.from_resolution(*res, scale=1.0)
    Vg, C = fc.vertices.shape[0], fc.indices.shape[0]
    if blob:
        sdf = (fc.vertices * torch.tensor([1.0, 1.2, 0.9])).norm(dim=-1, keepdim=True) - 0.6 + 0.05 * torch.randn(Vg, 1)
    else:
        sdf = torch.rand(Vg, 1) - 0.3
    verts = fc.vertices + 0.3 / max(res) * torch.tanh(torch.randn(Vg, 3))
    ins = dict(vertices=verts, sdf=sdf)
    if weights:
        ins.update(alpha=torch.randn(C, 8), beta=torch.randn(C, 12), gamma=torch.randn(C, 1))
    leaves = {k: v.clone().requires_grad_(True) for k, v in ins.items()}
    f2 = fc.replace(vertices=leaves["vertices"], sdf_values=leaves["sdf"],
                    **({k: leaves[k] for k in ("alpha", "beta", "gamma")} if weights else {}))
    mesh, L_dev = f2.dual_marching_cubes()
    ent = f2.compute_entropy()
    cv = torch.randn_like(mesh.vertices); cl = torch.randn_like(L_dev)
    ((mesh.vertices * cv).sum() + (L_dev * cl).sum() + 0.7 * ent).backward()
    out[f"{tag}.res"] = np.array(res)
    for k, v in ins.items():
        out[f"{tag}.in.{k}"] = v.numpy()
        out[f"{tag}.grad.{k}"] = leaves[k].grad.numpy()
    out[f"{tag}.out.vertices"] = mesh.vertices.detach().numpy(); out[f"{tag}.out.faces"] = mesh.indices.numpy()
    out[f"{tag}.out.L_dev"] = L_dev.detach().numpy(); out[f"{tag}.out.entropy"] = ent.detach().numpy()
    out[f"{tag}.cot.vertices"] = cv.numpy(); out[f"{tag}.cot.L_dev"] = cl.numpy()
    mv, mf, mL = O.extract(verts, sdf, res, ins.get("alpha"), ins.get("beta"), ins.get("gamma"))
    assert torch.equal(mf, mesh.indices) and torch.equal(mv, mesh.vertices.detach()) and torch.equal(mL, L_dev.detach())
    print(tag, res, "V", mv.shape[0], "F", mf.shape[0], "K", mL.shape[0], "restatement == reference")
np.savez_compressed(os.path.join(OUT, "ref_flexicubes.npz"), **out)
print("wrote", os.path.join(OUT, "ref_flexicubes.npz"))
