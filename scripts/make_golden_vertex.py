#!/usr/bin/env python3
"""Golden vectors for the 'vertex' sampling of the stage-1 loop and the trainer's schedule, produced by the reference's OWN code
(build container only; same stub recipe as scripts/make_golden.py):

    cd /tmp && PYTHONPATH=/tmp/stubs:/root/reference python /root/repo/scripts/make_golden_vertex.py

  * GaussianField.get_patches / get_gaussians_from_vertex (rfstudio/model/geosplat.py:520-620) on a bumpy icosphere, with the three
    hash encoders replaced by small fixed linear maps (the encoders themselves are pinned by ref_hashgrid.npz);
  * get_rotation_from_relative_vectors (rfstudio/graphics/math.py:159-188), incl. the near-opposite case;
  * GeoSplatTrainer.before_update (rfstudio/trainer/geosplat_trainer.py:209-266) at a handful of steps.
Writes tests/golden/ref_vertex.npz: inputs and outputs only.
"""
import os
import sys
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for name in ["open3d", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
             "cv2", "pyexr", "trimesh", "nvdiffrast", "nvdiffrast.torch", "kornia", "kornia.filters", "gsplat",
             "torchmetrics", "torchmetrics.functional", "torchmetrics.functional.image", "ffmpegcv", "nerfacc", "tyro",
             "skimage", "skimage.measure", "rfviser", "viser", "appdirs", "huggingface_hub", "pytorch3d",
             "pytorch3d.loss", "pytorch3d.structures", "rfstudio.graphics._mesh._optix", "rfstudio.graphics._mesh._splitsum",
             "tinycudann", "plotext", "imageio", "lpips", "matplotlib", "matplotlib.pyplot", "viser.transforms",
             "rfviser.transforms", "torchmetrics.image", "torchmetrics.image.lpip"]:
    sys.modules.setdefault(name, MagicMock())

import rfstudio.model.geosplat as GEO                                             # noqa: E402
import rfstudio.trainer.geosplat_trainer as TR                                    # noqa: E402
from rfstudio.graphics.math import get_rotation_from_relative_vectors            # noqa: E402

import geosplatting_amd.synthetic as syn                                          # noqa: E402

torch.manual_seed(5)
v, f = syn.icosphere(2)                                                           # 162 vertices, 320 faces
v = v * (0.7 + 0.1 * torch.sin(3 * v[:, :1]) * torch.cos(2 * v[:, 1:2]))
mesh = SimpleNamespace(normals=None, shape=(), num_faces=f.shape[0], num_vertices=v.shape[0], indices=f, vertices=v)
Wkd, Wks, Wz = torch.randn(3, 3) * 0.8, torch.randn(3, 2) * 0.8, torch.randn(3, 1) * 0.8
guess = torch.tensor([0.3, -0.2])
field = SimpleNamespace(kd_enc=lambda x: torch.sigmoid(x @ Wkd), ks_enc=lambda x: x @ Wks, z_enc=lambda x: x @ Wz, occ_enc=None,
                        device=torch.device("cpu"))
field.get_patches = lambda m: GEO.GaussianField.get_patches(field, m)
points, areas = field.get_patches(mesh)
splats, attrs = GEO.GaussianField.get_gaussians_from_vertex(field, 0.0, 0.0, 1.05, mesh, guess)
out = {"vertices": v.numpy(), "faces": f.numpy(), "scale": 1.05, "Wkd": Wkd.numpy(), "Wks": Wks.numpy(), "Wz": Wz.numpy(),
       "guess": guess.numpy(), "patch_normals": points.normals.numpy(), "patch_areas": areas.numpy(),
       "means": splats.means.numpy(), "scales": splats.scales.numpy(), "quats": splats.quats.numpy(),
       "opacities": splats.opacities.numpy(), "kd": attrs.kd.numpy(), "ks": attrs.ks.numpy(), "normals": attrs.normals.numpy()}

g = torch.Generator().manual_seed(3)
b = torch.randn(64, 3, generator=g)
b[0] = torch.tensor([0.0, 0.0, 1.0]); b[1] = torch.tensor([1.0, 0.0, 0.0]); b[2] = torch.tensor([0.3, -0.4, -0.86])
a = torch.tensor([0.0, 0.0, 1.0])
out.update({"rel_b": b.numpy(), "rel_rot": get_rotation_from_relative_vectors(a, b).numpy()})

tr = TR.GeoSplatTrainer()
steps = [0, 1, 25, 49, 50, 51, 199, 200, 250, 499, 500, 501, 1200]
keys = ["sample_method", "light_weight", "sdf_weight", "kd_grad_weight", "kd_regualr_perturb_std", "ks_grad_weight",
        "ks_regualr_perturb_std"]
rows = []
model = SimpleNamespace(sample_method="face", light_weight=0.0, sdf_weight=0.0, occ_weight=0.0, kd_grad_weight=0.0,
                        kd_regualr_perturb_std=0.0, ks_grad_weight=0.0, ks_regualr_perturb_std=0.0, normal_grad_weight=0.0)
for s in steps:
    tr.before_update(model, None, curr_step=s)
    rows.append([1.0 if model.sample_method == "vertex" else 0.0] + [float(getattr(model, k)) for k in keys[1:]])
out.update({"sched_steps": np.array(steps), "sched_keys": np.array(keys), "sched_values": np.array(rows, dtype=np.float64),
            "sched_defaults": np.array([tr.vertex_sample_warmup, tr.light_reg_begin, tr.light_reg_end, tr.light_reg_decay,
                                        tr.sdf_reg_begin, tr.sdf_reg_end, tr.sdf_reg_decay, tr.kd_grad_reg_begin, tr.kd_grad_reg_end,
                                        tr.kd_grad_reg_decay, tr.ks_grad_reg_begin, tr.ks_grad_reg_end, tr.ks_grad_reg_decay,
                                        tr.kd_regualr_perturb_std, tr.ks_regualr_perturb_std], dtype=np.float64)})
np.savez_compressed(os.path.join(OUT, "ref_vertex.npz"), **out)
print({k: (getattr(val, "shape", val)) for k, val in out.items()})
