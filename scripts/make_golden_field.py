#!/usr/bin/env python3
"""Generate tests/golden/ref_hashgrid.npz from the reference's OWN torch hash-encoding code (build container only):

    cd /tmp && PYTHONPATH=/tmp/stubs:/root/reference python /root/repo/scripts/make_golden_field.py

`HashEncoding.__setup__`, `hash_fn`, `pytorch_fwd` and `__call__` (rfstudio/model/components/encoding.py:124-241) and
`MLP.__call__` (rfstudio/nn/mlp.py:126-145) are executed as they are, on a plain namespace instead of the reference's
lazily-initialised Module object (its `ParameterModule.from_tensor` is replaced by a holder of the same tensor).
Configuration = GaussianField.kd_enc (rfstudio/model/geosplat.py:485-495) with a 2^12 table to keep the fixture small,
and the default 2^18 table for a second, smaller point set (only the touched table rows are stored).
Only input/output vectors are written.
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "tests", "golden")
for name in ["open3d", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
             "cv2", "pyexr", "trimesh", "nvdiffrast", "nvdiffrast.torch", "kornia", "kornia.filters", "gsplat",
             "torchmetrics", "torchmetrics.functional", "torchmetrics.functional.image", "ffmpegcv", "nerfacc", "tyro",
             "skimage", "skimage.measure", "rfviser", "viser", "appdirs", "huggingface_hub", "pytorch3d",
             "pytorch3d.loss", "pytorch3d.structures", "rfstudio.graphics._mesh._optix", "rfstudio.graphics._mesh._splitsum",
             "tinycudann", "plotext", "imageio", "lpips", "matplotlib", "matplotlib.pyplot", "viser.transforms",
             "rfviser.transforms", "torchmetrics.image", "torchmetrics.image.lpip"]:
    sys.modules.setdefault(name, MagicMock())

import rfstudio.model.components.encoding as ENC                         # noqa: E402
from rfstudio.nn.mlp import MLP                                           # noqa: E402

ENC.ParameterModule = types.SimpleNamespace(from_tensor=lambda t: types.SimpleNamespace(params=torch.nn.Parameter(t.clone())))
torch.manual_seed(5)


def make(log2, max_res, grad_scaling):
    ns = types.SimpleNamespace(num_levels=16, min_res=16, max_res=max_res, log2_hashmap_size=log2, features_per_level=2,
                               hash_init_scale=0.001, backend="torch", interpolation="linear", grad_scaling=grad_scaling)
    ENC.HashEncoding.__setup__(ns)
    ns.hash_fn = types.MethodType(ENC.HashEncoding.hash_fn, ns)
    ns.pytorch_fwd = types.MethodType(ENC.HashEncoding.pytorch_fwd, ns)
    return ns


def mlp_apply(weights, activation, feats):
    ns = types.SimpleNamespace(nn_layers=[types.SimpleNamespace(__call__=None)], skip_connection_set=set(), activation=activation,
                               initialize_weights=lambda d: None)
    layers = []
    for w in weights:
        lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False)
        lin.weight.data.copy_(w)
        layers.append(lin)
    ns.nn_layers = layers
    return MLP.__call__(ns, feats)


out = {}
for tag, log2, max_res, n in (("a", 12, 4096, 1000), ("b", 18, 4096, 200)):
    enc = make(log2, max_res, 16.0)
    with torch.no_grad():
        enc.hash_table.mul_(1000.0)                                       # O(1) entries: gradients are not drowned in 1e-3
    x = (torch.rand(n, 3) * 2 - 1)
    x[:8] = torch.tensor([[-1., -1., -1.], [1., 1., 1.], [0., 0., 0.], [0.5, -0.25, 0.125], [-1., 1., 0.], [0.999999, -0.999999, 0.3],
                          [0.25, 0.25, 0.25], [-0.5, 0.75, -0.125]])
    x.requires_grad_(True)
    feats = enc.pytorch_fwd(x)
    g = torch.randn_like(feats)
    (feats * g).sum().backward()
    touched = torch.nonzero(enc.hash_table.grad.abs().sum(-1) > 0).squeeze(-1)
    out.update({f"{tag}_log2": log2, f"{tag}_max_res": max_res, f"{tag}_scalings": enc.scalings.numpy(),
                f"{tag}_x": x.detach().numpy(), f"{tag}_feats": feats.detach().numpy(), f"{tag}_g": g.numpy(),
                f"{tag}_v_x": x.grad.numpy(), f"{tag}_touched": touched.numpy().astype(np.int64),
                f"{tag}_v_table_touched": enc.hash_table.grad[touched].numpy()})
    if tag == "a":
        out["a_table"] = enc.hash_table.detach().numpy()
    else:
        # a 2^18 x 16 table is 33 MB: store the seed instead and regenerate (torch.rand under manual_seed is stable)
        torch.manual_seed(77)
        tbl = (torch.rand(enc.hash_table.shape) * 2 - 1)
        with torch.no_grad():
            enc.hash_table.copy_(tbl)
        enc.hash_table.grad = None
        x2 = x.detach().clone().requires_grad_(True)
        feats = enc.pytorch_fwd(x2)
        (feats * g).sum().backward()
        touched = torch.nonzero(enc.hash_table.grad.abs().sum(-1) > 0).squeeze(-1)
        out.update({"b_table_seed": 77, "b_feats": feats.detach().numpy(), "b_v_x": x2.grad.numpy(),
                    "b_touched": touched.numpy().astype(np.int64), "b_v_table_touched": enc.hash_table.grad[touched].numpy()})

# the full encoder call of GaussianField.kd_enc: grad-scaling trick + MLP [32 -> 32 -> 32 -> 3], sigmoid, no bias
enc = make(12, 4096, 16.0)
with torch.no_grad():
    enc.hash_table.copy_(torch.tensor(out["a_table"]))
ws = [torch.randn(32, 32) * 0.3, torch.randn(32, 32) * 0.3, torch.randn(3, 32) * 0.3]
enc.mlp = lambda f: mlp_apply(ws, "sigmoid", f)
x = torch.tensor(out["a_x"][:256]).requires_grad_(True)
y = ENC.HashEncoding.__call__(enc, x)
gy = torch.randn_like(y)
(y * gy).sum().backward()
touched = torch.nonzero(enc.hash_table.grad.abs().sum(-1) > 0).squeeze(-1)
out.update({"c_w0": ws[0].numpy(), "c_w1": ws[1].numpy(), "c_w2": ws[2].numpy(), "c_y": y.detach().numpy(), "c_gy": gy.numpy(),
            "c_v_x": x.grad.numpy(), "c_touched": touched.numpy().astype(np.int64),
            "c_v_table_touched": enc.hash_table.grad[touched].numpy()})
np.savez_compressed(os.path.join(OUT, "ref_hashgrid.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
