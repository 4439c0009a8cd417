"""The render-and-backward step of the path for a batch of views -- the counterpart of
``GeoSplatter.render_report`` (rfstudio/model/geosplat.py:856-927: prefilter once, then a loop of
``attrs.splat`` over the views) followed by the autograd backward the trainer triggers
(rfstudio/optim/optimizer.py:107), with the data-parallel gradient sum of geosplatting_amd.parallel.

Order of work per step (all on the current HIP stream, kernels in libgeosplat_hip.so):
    as_splitsum(cubemap)                                   S5 forward   (once per step)
    for each local view:  shade -> rasterize -> tone-map    S1-S4, A1-A5
                          backward of the same              A6, A7, S1-S3 backward
    prefilter backward with the texel gradients summed over the views   S5 backward (once per step)
    flat-bucket all-reduce over ranks                       RCCL
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import torch
from torch import Tensor

from .cameras import Camera
from .parallel import GradBucket
from .shading import RenderableAttrs
from .splitsum import TextureSplitSum, as_splitsum
from .synthetic import SplatSet

PARAM_NAMES = ("means", "scales", "quats", "opacities", "normals", "kd", "ks", "cubemap", "exposure")


@dataclass
class PathParams:
    """Leaf tensors of the path (replicated on every rank)."""
    means: Tensor; scales: Tensor; quats: Tensor; opacities: Tensor       # raw Splats fields
    normals: Tensor; kd: Tensor; ks: Tensor                                 # RenderableAttrs
    cubemap: Tensor                                                         # [6,R,R,3]
    exposure: Tensor                                                        # scalar (already exp'ed)

    def named(self) -> Dict[str, Tensor]:
        return {k: getattr(self, k) for k in PARAM_NAMES}

    def shapes(self) -> Dict[str, Sequence[int]]:
        return {k: tuple(v.shape) if v.dim() > 0 else (1,) for k, v in self.named().items()}


class RenderStep:
    def __init__(self, params: PathParams, min_roughness: float = 0.1, max_metallic: float = 1.0, mode: str = "pbr",
                 tone_type: str = "naive", prefilter: bool = True):
        self.p = params
        self.min_roughness, self.max_metallic, self.mode, self.tone_type = min_roughness, max_metallic, mode, tone_type
        self.prefilter = prefilter
        self.bucket = GradBucket(params.shapes(), params.means.device)
        self._static_env: Optional[TextureSplitSum] = None

    def __call__(self, cameras: List[Camera], upstream: Callable[[int, Tensor], Tensor], all_reduce: bool = True,
                 keep_images: bool = False):
        """Forward + backward for `cameras`; `upstream(i, image)` returns d(loss)/d(image) for local view i.
        Returns (grads dict of views into the flat bucket, images or None)."""
        p = self.p
        leaves = {k: v.detach().requires_grad_(True) for k, v in p.named().items()}
        if self.prefilter:
            env = as_splitsum(leaves["cubemap"])
        else:
            if self._static_env is None:
                with torch.no_grad():
                    self._static_env = as_splitsum(p.cubemap)
            env = self._static_env
        # cut the graph at the pyramid so that the prefilter backward runs ONCE with the summed texel gradients
        base_leaf = env.base.detach().requires_grad_(True)
        level_leaves = [l.detach().requires_grad_(True) for l in env.levels]
        env_leaf = TextureSplitSum(base_leaf, level_leaves, env.min_roughness, env.max_roughness)

        class _G:  # raw Splats-like view
            means = leaves["means"]; scales = leaves["scales"]; quats = leaves["quats"]; opacities = leaves["opacities"]
        attrs = RenderableAttrs(kd=leaves["kd"], ks=leaves["ks"], normals=leaves["normals"])
        images = []
        for i, cam in enumerate(cameras):
            img = attrs.splat(_G, [cam], exposure=leaves["exposure"], envmap=env_leaf, min_roughness=self.min_roughness,
                              max_metallic=self.max_metallic, mode=self.mode, tone_type=self.tone_type)
            img.backward(upstream(i, img.detach()))
            if keep_images:
                images.append(img.detach())
        if self.prefilter:
            pairs = [(env.base, base_leaf.grad)] + [(l, ll.grad) for l, ll in zip(env.levels, level_leaves)]
            outs = [o for o, g in pairs if g is not None]
            gouts = [g for o, g in pairs if g is not None]
            if outs:
                torch.autograd.backward(outs, gouts)
        grads = {k: leaves[k].grad for k in PARAM_NAMES}
        self.bucket.pack(grads)
        if all_reduce:
            self.bucket.all_reduce()
        return self.bucket.unpack(), (images if keep_images else None)


def params_from_scene(scene, device, exposure: float = 1.0) -> PathParams:
    sp: SplatSet = scene.splats
    d = lambda t: t.to(device).contiguous()
    return PathParams(d(sp.means), d(sp.scales), d(sp.quats), d(sp.opacities), d(scene.normals), d(scene.kd),
                      d(scene.ks), d(scene.cubemap), torch.tensor(exposure, dtype=torch.float32, device=device))
