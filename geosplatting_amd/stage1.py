"""The reference's stage-1 model and trainer step (BASELINE config 5: "full geosplat.py MGAdaptor training loop,
FlexiCubes -> splats -> PBR"), composed from this package's HIP-backed pieces and data-parallel over views.

`Stage1Model` mirrors `GeoSplatter` (rfstudio/model/geosplat.py:676-927): parameters `sdf_params`, `deform_params`,
`weight_params`, `cubemap`, `exposure_params`, the `GaussianField` encoders, the scalar weights of `__setup__`
(:703-728) and `get_geometry / get_envmap / get_gsplat('face') / render_report` in the reference's order.  `train_step`
is `GeoSplatTrainer.step` (rfstudio/trainer/geosplat_trainer.py:150-186): per-view random-background SSIM/L1 + mask loss,
mean over the views, plus the regulariser.

Data parallel (one process per GPU): every rank holds the same parameters, extracts the same mesh, renders ITS views
(`views[rank::world_size]`), and the gradients of all parameters travel as ONE flat fp32 all-reduce (RCCL over xGMI).
With the loss normalised by the global view count and the regulariser by the world size, the reduced gradient is the
single-process gradient.  `smooth_type` 'jitter' (the reference's default) runs on the fused engine too; the 'grad' / 'tv'
branches and `normal_grad_weight` of render_report (:881-922: extra un-shaded renders of kd / ks / normals, kornia's
`spatial_gradient` restated below -- kornia is absent, so that one function is unpinned) run on the autograd step.  The stage hand-off files are: `export_model`
(:839-854, the dict stage 2 loads), `state_dict` with the reference's names and `<step:010d>.ckpt` checkpoints
(rfstudio/engine/train.py:172-190).
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from .cameras import Camera
from .field import GaussianField
from .flexicubes import FlexiCubes, get_geometry
from .loss import photo_loss
from .splitsum import as_splitsum

def spatial_gradient(img: Tensor) -> Tensor:
    """kornia.filters.spatial_gradient(x[None].permute(0,3,1,2), order=1)[0] for an [H,W,C] image -> [C,2,H,W]: normalised Sobel
    pair (kernel / 8), replicate padding, cross-correlation; channel 0 = d/dx, 1 = d/dy (kornia ~= 0.7 defaults: mode='sobel',
    normalized=True).  Restated from the published definition."""
    x = img.permute(2, 0, 1)[:, None]                                     # [C,1,H,W]
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], device=img.device, dtype=img.dtype) / 8.0
    k = torch.stack((kx, kx.t()))[:, None]                                # [2,1,3,3]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1), mode="replicate")
    return torch.nn.functional.conv2d(xp, k)                              # [C,2,H,W]


def _edge_aware(rendered: Tensor, gt_rgb: Tensor) -> Tensor:
    """first_order_edge_aware_loss of rfstudio/model/geosplat.py:885-889"""
    return (spatial_gradient(rendered).abs() * (-spatial_gradient(gt_rgb).abs()).exp()).sum(1).mean()


def _tv(rendered: Tensor) -> Tensor:
    """tv_loss of rfstudio/model/geosplat.py:907-910"""
    return (rendered[1:, :] - rendered[:-1, :]).square().mean() + (rendered[:, 1:] - rendered[:, :-1]).square().mean()


_INITIAL_GUESS = {"outdoor": (0.0, 0.0), "diffuse": (0.0, -3.0), "hybrid": (-3.0, -3.0), "specular": (-3.0, 0.0),
                  "glossy": (-3.0, 0.0)}                                   # geosplat.py:729-740


_allocator_configured = False


def configure_allocator() -> None:
    """The stage-1 loop extracts a different mesh -- a different number of Gaussians -- in every iteration, so every per-Gaussian
    buffer has a size PyTorch's caching allocator has never seen: it answered each with a fresh hipMalloc and kept the old blocks
    (scripts/soak_stage1.py, 128^3 grid: 253 GiB reserved after 100 iterations for 8.5 GiB of live memory, 77-160 ms per iteration).
    Rounding request sizes up to an eighth of a power of two makes the sizes repeat: 26 GiB flat, 40 ms per iteration.  Applied once,
    unless the user configured the allocator through PyTorch's own environment variables (PYTORCH_HIP_ALLOC_CONF / PYTORCH_ALLOC_CONF)."""
    global _allocator_configured
    if _allocator_configured:
        return
    _allocator_configured = True
    if os.environ.get("PYTORCH_HIP_ALLOC_CONF") or os.environ.get("PYTORCH_CUDA_ALLOC_CONF") or os.environ.get("PYTORCH_ALLOC_CONF"):
        return
    conf = "roundup_power2_divisions:8"
    setter = getattr(torch._C, "_accelerator_setAllocatorSettings", None) or getattr(torch.cuda.memory, "_set_allocator_settings", None)
    if setter is not None:
        try:
            setter(conf)
            # process-wide and therefore said out loud: every allocation of the host application is
            # rounded up to an eighth of a power of two (<= 12.5 % internal fragmentation) from here on
            print(f"[geosplatting_amd.stage1] caching allocator set to '{conf}' for the whole process "
                  f"(set PYTORCH_HIP_ALLOC_CONF yourself to keep your own)", file=sys.stderr)
        except Exception as e:                               # an allocator without this knob: nothing lost but the optimisation
            print(f"[geosplatting_amd.stage1] allocator setting '{conf}' not applied: {e!r}", file=sys.stderr)


class Stage1Model:
    def __init__(self, resolution: int = 32, *, scale: float = 1.05, light_resolution: int = 512, min_roughness: float = 0.1,
                 max_metallic: float = 1.0, initial_guess: str = "hybrid", device="cuda", seed: int = 0,
                 log2_hashmap_size: int = 18, sdf_init: Optional[Tensor] = None):
        dev = torch.device(device)
        if dev.type == "cuda":
            configure_allocator()
        g = torch.Generator(device=dev).manual_seed(seed)
        self.resolution, self.scale = resolution, scale
        self.min_roughness, self.max_metallic = min_roughness, max_metallic
        self.grid = FlexiCubes.from_resolution(resolution, device=dev, random_sdf=False, scale=scale)
        V, Cn = self.grid.vertices.shape[0], resolution ** 3
        # from_resolution's sdf = U(0,1) - 0.1 (:447-451), replicated on every rank from the seed
        sdf0 = (torch.rand(V, 1, device=dev, generator=g) - 0.1) if sdf_init is None else sdf_init.to(dev).reshape(V, 1).clone()
        self.sdf_params = sdf0.requires_grad_(True)
        self.deform_params = torch.zeros(V, 3, device=dev, requires_grad=True)
        self.weight_params = torch.zeros(Cn, 21, device=dev, requires_grad=True)
        self.cubemap = torch.full((6, light_resolution, light_resolution, 3), 0.5, device=dev, requires_grad=True)   # :741-748
        self.exposure_params = torch.zeros(1, device=dev, requires_grad=True)
        self.field = GaussianField(device=dev, log2_hashmap_size=log2_hashmap_size, seed=seed + 1)
        self.initial_guess_bias = torch.tensor(_INITIAL_GUESS[initial_guess], device=dev)
        # scalar weights of __setup__ (:715-727); the trainer's schedule sets them
        self.sdf_weight = 0.0; self.light_weight = 0.0
        self.kd_grad_weight = 0.0; self.kd_regualr_perturb_std = 0.0
        self.ks_grad_weight = 0.0; self.ks_regualr_perturb_std = 0.0
        self.normal_grad_weight = 0.0
        self.smooth_type = "jitter"                              # 'jitter' | 'grad' | 'tv'  (geosplat.py:697)
        self.sample_method = "face"                              # 'face' | 'vertex': set by the trainer's schedule (GeoSplatSchedule)
        self.background_color = torch.ones(3, device=dev)        # get_background_color() of a 'white' model
        self.last_num_gaussians = 0
        # random streams owned by the model: the kd / ks jitter must be IDENTICAL on every rank (replicas extract and perturb
        # the same Gaussians), the training-background noise differs per rank and advances every iteration
        self._jitter_gen = torch.Generator(device=dev).manual_seed(seed * 7919 + 17)
        self._bg_gens: Dict[int, torch.Generator] = {}
        self._dev, self._seed = dev, seed

    def bg_generator(self, rank: int) -> torch.Generator:
        """persistent per-rank stream of the trainer's random training background (geosplat_trainer.py:172)"""
        if rank not in self._bg_gens:
            self._bg_gens[rank] = torch.Generator(device=self._dev).manual_seed((self._seed * 1000003 + rank) * 2654435761 % (2 ** 63))
        return self._bg_gens[rank]

    # ------------------------------------------------------------------------------------------------- parameters
    def named_parameters(self) -> Dict[str, Tensor]:
        out = {"sdf_params": self.sdf_params, "deform_params": self.deform_params, "weight_params": self.weight_params,
               "cubemap": self.cubemap, "exposure_params": self.exposure_params}
        for name, enc in (("kd_enc", self.field.kd_enc), ("ks_enc", self.field.ks_enc), ("z_enc", self.field.z_enc)):
            for i, p in enumerate(enc.parameters()):
                out[f"field.{name}.{i}"] = p
        return out

    def parameters(self) -> List[Tensor]:
        return list(self.named_parameters().values())

    # ------------------------------------------------------------------------------------------------- stage hand-off
    def export_model(self, path) -> None:
        """`GeoSplatter.export_model` (rfstudio/model/geosplat.py:839-854): the file stage 2 starts from
        (`GeoSplatterMC.__setup__` reads exactly these keys, rfstudio/model/geosplat_mc.py:56-70)."""
        with torch.no_grad():
            attributes = {
                "geom_scale": self.scale,
                "resolution": self.resolution,
                "min_roughness": self.min_roughness,
                "max_metallic": self.max_metallic,
                "exposure": self.exposure_params.detach().cpu().clone(),
                "cubemap": self.cubemap.detach().cpu().clone(),
                "deforms": self.deform_params.detach().cpu().clone(),
                "weights": self.weight_params.detach().cpu().clone(),
                "sdfs": self.sdf_params.detach().cpu().clone(),
                "ks_enc": self.field.ks_enc.state_dict(),
                "initial_guess": self.initial_guess_bias.detach().cpu().clone(),
            }
        torch.save(attributes, path)

    @classmethod
    def from_export(cls, path, device="cuda", **kwargs) -> "Stage1Model":
        """Rebuild the geometry / lighting / ks side of a model from an `export_model` file, as the stage-2 loader does
        (geosplat_mc.py:56-70); the kd / z encoders are not part of the hand-off and start fresh."""
        a = torch.load(path, map_location="cpu")
        log2 = int(round(float(torch.log2(torch.tensor(a["ks_enc"]["encoder.params"].shape[0] / 16.0)))))
        m = cls(int(a["resolution"]), scale=float(a["geom_scale"]), light_resolution=int(a["cubemap"].shape[1]),
                min_roughness=float(a["min_roughness"]), max_metallic=float(a["max_metallic"]), device=device,
                log2_hashmap_size=log2, **kwargs)
        with torch.no_grad():
            m.exposure_params.copy_(a["exposure"].to(m.exposure_params.device))
            m.cubemap.copy_(a["cubemap"].to(m.cubemap.device))
            m.deform_params.copy_(a["deforms"].to(m.deform_params.device))
            m.weight_params.copy_(a["weights"].to(m.weight_params.device))
            m.sdf_params.copy_(a["sdfs"].to(m.sdf_params.device))
            m.initial_guess_bias = a["initial_guess"].to(m.initial_guess_bias.device).float()
        m.field.ks_enc.load_state_dict(a["ks_enc"])
        return m

    def state_dict(self) -> Dict[str, Tensor]:
        """Names of the reference's `GeoSplatter.state_dict()` (probed in the build container): the five parameters,
        `initial_guess_bias`, the unused `latlng` placeholder, and the three encoders under `field.<name>.`."""
        sd = {"exposure_params": self.exposure_params, "deform_params": self.deform_params, "sdf_params": self.sdf_params,
              "weight_params": self.weight_params, "initial_guess_bias": self.initial_guess_bias, "cubemap": self.cubemap,
              "latlng": torch.zeros(256, 512, 3)}
        out = {k: v.detach().cpu().clone() for k, v in sd.items()}
        for name in ("kd_enc", "ks_enc", "z_enc"):
            for k, v in getattr(self.field, name).state_dict().items():
                out[f"field.{name}.{k}"] = v
        return out

    def load_state_dict(self, sd: Dict[str, Tensor]) -> None:
        with torch.no_grad():
            for k in ("exposure_params", "deform_params", "sdf_params", "weight_params", "cubemap"):
                getattr(self, k).copy_(sd[k].to(getattr(self, k).device))
            self.initial_guess_bias = sd["initial_guess_bias"].to(self.initial_guess_bias.device).float()
        for name in ("kd_enc", "ks_enc", "z_enc"):
            pre = f"field.{name}."
            getattr(self.field, name).load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})

    def save_checkpoint(self, ckpt_dir, step: int) -> str:
        """`TrainTask.save_checkpoint` (rfstudio/engine/train.py:172-175): `<ckpt_dir>/<step:010d>.ckpt` = state_dict."""
        os.makedirs(ckpt_dir, exist_ok=True)
        path = os.path.join(ckpt_dir, f"{step:010d}.ckpt")
        torch.save(self.state_dict(), path)
        return path

    def load_checkpoint(self, ckpt_dir, step: Optional[int] = None) -> Optional[int]:
        """`TrainTask.load_checkpoint` (train.py:177-190): the given step, else the newest `*.ckpt`; None if there is none."""
        if not os.path.isdir(ckpt_dir):
            return None
        if step is None:
            steps = [int(f.rsplit(".", 1)[0]) for f in os.listdir(ckpt_dir) if f.endswith(".ckpt")]
            if not steps:
                return None
            step = max(steps)
        path = os.path.join(ckpt_dir, f"{step:010d}.ckpt")
        if not os.path.exists(path):
            return None
        self.load_state_dict(torch.load(path, map_location="cpu"))
        return step

    # ------------------------------------------------------------------------------------------------- forward pieces
    def get_geometry(self) -> Tuple[Tuple[Tensor, Tensor], Tensor]:
        """:751-769"""
        return get_geometry(self.grid, self.deform_params, self.sdf_params, self.weight_params, scale=self.scale,
                            resolution=self.resolution, sdf_weight=self.sdf_weight)

    def get_envmap(self):
        """:780-785 -> (TextureSplitSum, white-balance regulariser)"""
        white = self.cubemap.mean(-1, keepdim=True)
        return as_splitsum(self.cubemap), (self.cubemap - white).abs().mean()

    def get_gsplat(self, sampling: Optional[str] = None):
        """:787-831 -> (mesh, splats, attrs, regularisation).  sampling 'face' (six Gaussians per face through the MGAdapter) or
        'vertex' (one per vertex: what the trainer selects for its first `vertex_sample_warmup` steps); default: self.sample_method."""
        sampling = self.sample_method if sampling is None else sampling
        if sampling not in ("face", "vertex"):
            raise ValueError(sampling)
        (v, f), reg = self.get_geometry()
        self.last_num_gaussians = f.shape[0] * 6 if sampling == "face" else v.shape[0]
        # the jittered encoder evaluations and their L1 terms exist for smooth_type == 'jitter' only (:800-801): under 'grad' /
        # 'tv' the perturbation stds are zeroed, nothing is drawn from the jitter generator and no jitter term enters the loss
        kd_std = self.kd_regualr_perturb_std if self.smooth_type == "jitter" else 0.0
        ks_std = self.ks_regualr_perturb_std if self.smooth_type == "jitter" else 0.0
        if sampling == "face":
            splats, attrs, _ = self.field.get_gaussians_from_face(v, f, kd_std, ks_std, scale=self.scale,
                                                                  initial_guess=self.initial_guess_bias, generator=self._jitter_gen)
        else:
            splats, attrs = self.field.get_gaussians_from_vertex(v, f, kd_std, ks_std, scale=self.scale,
                                                                 initial_guess=self.initial_guess_bias, generator=self._jitter_gen)
        if kd_std > 0 and self.kd_grad_weight > 0:
            reg = reg + self.kd_grad_weight * (attrs.kd_jitter - attrs.kd).abs().mean()
        if ks_std > 0 and self.ks_grad_weight > 0:
            reg = reg + self.ks_grad_weight * (attrs.ks_jitter - attrs.ks).abs().mean()
        return (v, f), splats, attrs, reg

    def smoothing_regularization(self, splats, attrs, cameras: Sequence[Camera], gt_rgba: Optional[Sequence[Tensor]],
                                 batch_size: int) -> Tensor:
        """The 'grad' / 'tv' branches and `normal_grad_weight` of render_report (:881-922): edge-aware first-order / total-variation
        penalties on UN-shaded renders of kd, (0, ks) and normals * 0.5 + 0.5 over the model's background colour."""
        from .shading import render_rgb
        reg = splats.means.new_zeros(())
        need_gt = (self.smooth_type == "grad" and (self.kd_grad_weight > 0 or self.ks_grad_weight > 0)) or self.normal_grad_weight > 0
        if need_gt and gt_rgba is None:
            raise ValueError("smooth_type='grad' / normal_grad_weight need the ground-truth images (gt_outputs.blend(background))")
        bg = self.background_color
        rr = lambda colors, cam: render_rgb(splats.means, splats.scales, splats.quats, splats.opacities, colors, cam, bg)
        gts = None if not need_gt else [g[..., :3] * g[..., 3:] + bg * (1 - g[..., 3:]) for g in gt_rgba]   # RGBAImages.blend
        ks3 = torch.cat((torch.zeros_like(attrs.ks[..., :1]), attrs.ks), dim=-1)
        for i, cam in enumerate(cameras):
            if self.smooth_type == "grad":
                if self.kd_grad_weight > 0:
                    reg = reg + _edge_aware(rr(attrs.kd, cam), gts[i]) * self.kd_grad_weight / batch_size
                if self.ks_grad_weight > 0:
                    reg = reg + _edge_aware(rr(ks3, cam), gts[i]) * self.ks_grad_weight / batch_size
            if self.normal_grad_weight > 0:
                reg = reg + _edge_aware(rr(attrs.normals * 0.5 + 0.5, cam), gts[i]) * self.normal_grad_weight / batch_size
            if self.smooth_type == "tv":
                if self.kd_grad_weight > 0:
                    reg = reg + _tv(rr(attrs.kd, cam)) * self.kd_grad_weight / batch_size
                if self.ks_grad_weight > 0:
                    reg = reg + _tv(rr(ks3, cam)) * self.ks_grad_weight / batch_size
        return reg

    def render_report(self, cameras: Sequence[Camera], gt_rgba: Optional[Sequence[Tensor]] = None,
                      batch_size: Optional[int] = None) -> Tuple[List[Tensor], int, Tensor]:
        """:856-927 -> (tone-mapped linear RGBA image per camera, #Gaussians, regularisation); gt_rgba (linear RGBA per camera) is
        only read by the 'grad' / normal smoothing branches; batch_size = number of views of the whole step (all ranks)."""
        _, splats, attrs, reg = self.get_gsplat()
        envmap, light_reg = self.get_envmap()
        exposure = self.exposure_params.exp()[0]
        images = [attrs.splat(splats, [cam], exposure=exposure, envmap=envmap, min_roughness=self.min_roughness,
                              max_metallic=self.max_metallic).reshape(cam.height, cam.width, 4) for cam in cameras]
        self._last_smoothing = reg.new_zeros(())
        if self.smooth_type != "jitter" or self.normal_grad_weight > 0:
            self._last_smoothing = self.smoothing_regularization(splats, attrs, cameras, gt_rgba, batch_size or len(cameras))
            reg = reg + self._last_smoothing                 # per-view terms (already / batch_size): NOT replicated across ranks
        return images, splats.means.shape[0], reg + light_reg * self.light_weight


@dataclass
class GeoSplatSchedule:
    """The per-step schedule of the reference's stage-1 trainer (GeoSplatTrainer.before_update / after_update / the gradient hooks
    of setup, rfstudio/trainer/geosplat_trainer.py:20-62,66-69,209-266) as plain host code: which sampling the model uses, the
    linear ramps of its regularisation weights, the x64 scaling of the environment-map gradient and the floor of the cubemap."""
    vertex_sample_warmup: int = 50
    light_reg_begin: float = 2e-3
    light_reg_end: float = 2e-3
    light_reg_decay: int = 500
    sdf_reg_begin: float = 0.2
    sdf_reg_end: float = 0.12
    sdf_reg_decay: int = 500
    kd_grad_reg_begin: float = 0.0
    kd_grad_reg_end: float = 0.03
    kd_grad_reg_decay: int = 500
    kd_regualr_perturb_std: float = 0.01
    ks_grad_reg_begin: float = 0.0
    ks_grad_reg_end: float = 0.001
    ks_grad_reg_decay: int = 500
    ks_regualr_perturb_std: float = 0.01
    normal_grad_reg_begin: float = 0.0
    normal_grad_reg_end: float = 0.5
    normal_grad_reg_decay: int = 0
    light_gradient_scale: float = 64.0
    cubemap_floor: float = 1e-2

    @staticmethod
    def _ramp(begin: float, end: float, t: float) -> float:
        return begin - (begin - end) * min(1.0, t)

    def before_update(self, model, curr_step: int) -> None:
        model.sample_method = "vertex" if (self.vertex_sample_warmup > 0 and curr_step < self.vertex_sample_warmup) else "face"
        model.light_weight = self._ramp(self.light_reg_begin, self.light_reg_end, curr_step / self.light_reg_decay)
        if self.sdf_reg_decay > 0:
            model.sdf_weight = self._ramp(self.sdf_reg_begin, self.sdf_reg_end, curr_step / self.sdf_reg_decay)
        if self.kd_grad_reg_decay > 0:
            model.kd_grad_weight = self._ramp(self.kd_grad_reg_begin, self.kd_grad_reg_end, curr_step / self.kd_grad_reg_decay)
            model.kd_regualr_perturb_std = self.kd_regualr_perturb_std
        if self.ks_grad_reg_decay > 0:
            model.ks_grad_weight = self._ramp(self.ks_grad_reg_begin, self.ks_grad_reg_end, curr_step / self.ks_grad_reg_decay)
            model.ks_regualr_perturb_std = self.ks_regualr_perturb_std
        if self.normal_grad_reg_decay > 0:
            model.normal_grad_weight = self._ramp(self.normal_grad_reg_begin, self.normal_grad_reg_end,
                                                  max(curr_step - 200, 0) / self.normal_grad_reg_decay)

    def scale_light_gradient(self, model) -> None:
        """setup's `model.cubemap.register_hook(lambda grad: grad * 64)`, applied to the finished gradient (the fused step writes
        .grad directly, so a tensor hook would not see it)"""
        if model.cubemap.grad is not None:
            model.cubemap.grad.mul_(self.light_gradient_scale)

    def after_update(self, model, curr_step: int) -> None:
        with torch.no_grad():
            model.cubemap.clamp_min_(self.cubemap_floor)


def train_step_fused(model: Stage1Model, cameras: Sequence[Camera], gt_rgba: Sequence[Tensor], *, gt_is_srgb: bool = True,
                     use_mask_loss: bool = True, rank: int = 0, world_size: int = 1, seed: int = 0,
                     train_bg: Optional[Sequence[Tensor]] = None) -> Dict[str, Tensor]:
    """`train_step` with the render / loss half on the hand-scheduled engine (engine.RenderStep: C-ABI drivers on three
    streams, loss gradient from gs_photo_loss, per-Gaussian gradient bucket) instead of one autograd graph per view.
    The all-reduce happens at the per-Gaussian cut (every rank extracted the same Gaussians), after which each rank
    runs the same field / MGAdapter / FlexiCubes backward: no second collective.  Replicas must start identical
    (`broadcast_parameters`) and stay so: the backward kernels behind the cut use fp32 atomics, so callers re-broadcast
    every few hundred steps (`_main`: every 200); the Gaussian count is checked across ranks before every
    collective (a topology that differs on one rank would otherwise reduce buffers of different sizes)."""
    from .engine import PathParams, RenderStep
    from .loss import TrainerUpstream
    n_total = len(cameras)
    mine = list(range(rank, n_total, world_size))
    if model.smooth_type != "jitter" or model.normal_grad_weight > 0:
        raise NotImplementedError("the 'grad' / 'tv' / normal smoothing renders run on the autograd step (train_step)")
    params = model.parameters()
    for p in params:
        p.grad = None
    _, splats, attrs, reg = model.get_gsplat()
    white = model.cubemap.mean(-1, keepdim=True)
    reg = reg + (model.cubemap - white).abs().mean() * model.light_weight
    exposure = model.exposure_params.exp()
    cut = [splats.means, splats.scales, splats.quats, splats.opacities, attrs.normals, attrs.kd, attrs.ks]
    if world_size > 1:
        assert_same_count(splats.means.shape[0])
    pp = PathParams(*[t.detach().contiguous() for t in cut], model.cubemap.detach(), exposure.detach().reshape(()))
    step = getattr(model, "_render_step", None)
    if step is None:
        step = model._render_step = RenderStep(pp, min_roughness=model.min_roughness, max_metallic=model.max_metallic)
    else:
        step.rebind(pp)
    up = TrainerUpstream([gt_rgba[i] for i in mine], n_total, gt_is_srgb=gt_is_srgb, use_mask_loss=use_mask_loss,
                         train_bg=None if train_bg is None else [train_bg[i] for i in mine],
                         generator=model.bg_generator(rank), device=splats.means.device)   # `mine` may be empty (world > views)
    # capacity protocol (engine.RenderStep.poll_capacity): the step runs without (V, I) read-backs; a view that outgrew the
    # intersection capacity (the extracted surface changes every iteration) is reported afterwards and the step repeated with
    # the raised capacity -- collectively, since the per-Gaussian gradients were already reduced over the ranks
    for attempt in range(4):
        g, _ = step([cameras[i] for i in mine], up, all_reduce=world_size > 1)
        ok = step.poll_capacity(wait=True)
        if world_size > 1:
            flag = torch.tensor([0.0 if ok else 1.0], device=splats.means.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            ok = float(flag.item()) == 0.0
        if ok:
            break
        up = TrainerUpstream([gt_rgba[i] for i in mine], n_total, gt_is_srgb=gt_is_srgb, use_mask_loss=use_mask_loss,
                             train_bg=None if train_bg is None else [train_bg[i] for i in mine],
                             generator=model.bg_generator(rank), device=splats.means.device)
    else:
        raise RuntimeError("intersection capacity still exceeded after 4 attempts")
    heads = cut + [exposure]
    gh = [g["means"], g["scales"], g["quats"], g["opacities"], g["normals"], g["kd"], g["ks"], g["exposure"].reshape(1)]
    keep = [(h, gg) for h, gg in zip(heads, gh) if h.requires_grad]
    torch.autograd.backward([h for h, _ in keep] + [reg], [gg.reshape(h.shape) for h, gg in keep] + [torch.ones_like(reg)])
    model.cubemap.grad = g["cubemap"].clone() if model.cubemap.grad is None else model.cubemap.grad + g["cubemap"]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    return {"loss_local_views": up.mean_loss() * (n_total / max(1, len(mine))), "regularization": reg.detach(),
            "#gaussians": torch.tensor(splats.means.shape[0]), "exposure": exposure.detach().mean()}


def broadcast_parameters(model: "Stage1Model", src: int = 0, group=None) -> None:
    """Make every replica bit-identical to rank `src` (start-up, and periodically against low-bit drift of the
    atomics-based backward kernels): one flat broadcast."""
    params = model.parameters()
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        dist.broadcast(flat, src=src, group=group)
        o = 0
        for p in params:
            p.copy_(flat[o:o + p.numel()].view_as(p)); o += p.numel()


def assert_same_count(n: int, group=None) -> None:
    """Every rank must have extracted the same number of Gaussians before a flat collective over per-Gaussian buffers."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([n, -n], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    lo, hi = -int(t[1]), int(t[0])
    if lo != hi:
        raise RuntimeError(f"replicas diverged: this rank extracted {n} Gaussians, the group has between {lo} and {hi}; "
                           "call broadcast_parameters() (and keep the jitter generator in step on every rank)")


def flat_all_reduce(grads: List[Tensor], group=None) -> None:
    """One flat fp32 all-reduce (sum) over a list of gradient tensors, written back in place."""
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g)); o += g.numel()


def train_step(model: Stage1Model, cameras: Sequence[Camera], gt_rgba: Sequence[Tensor], *, gt_is_srgb: bool = True,
               use_mask_loss: bool = True, rank: int = 0, world_size: int = 1, all_reduce: Optional[bool] = None,
               bg_generator: Optional[torch.Generator] = None, train_bg: Optional[Sequence[Tensor]] = None) -> Dict[str, Tensor]:
    """One `GeoSplatTrainer.step` + backward over ALL `cameras`, of which this rank renders `cameras[rank::world_size]`.
    Leaves d(loss + regularisation)/d(parameter) of the GLOBAL batch in every parameter's .grad (after the flat
    all-reduce when world_size > 1).  train_bg: fixed per-view backgrounds instead of the trainer's torch.rand_like
    (tests).  Returns detached metrics of the local views."""
    from .viewbatch import retry_on_capacity
    n_total = len(cameras)
    mine = list(range(rank, n_total, world_size))

    def attempt():
        # (a view that outgrew the capacity the earlier steps taught raises GeoSplatCapacityError from backward(), before anything
        #  consumed the step: the capacity has been raised, the step is simply repeated -- viewbatch.retry_on_capacity)
        for p in model.parameters():
            p.grad = None
        images, num_gaussians, reg = model.render_report([cameras[i] for i in mine], [gt_rgba[i] for i in mine], batch_size=n_total)
        smooth = model._last_smoothing                                  # per-view smoothing terms of THIS rank's views
        total = (reg - smooth) / world_size + smooth                    # the rest is identical on every rank: counted once
        local = []
        for i, img in zip(mine, images):
            gt = gt_rgba[i]
            bg = train_bg[i] if train_bg is not None else \
                torch.rand(img.shape[0], img.shape[1], 3, device=img.device, generator=bg_generator)    # :172
            loss, _ = photo_loss(img[..., :3], img[..., 3:], gt, bg, gt_is_srgb=gt_is_srgb, use_mask_loss=use_mask_loss)
            local.append(loss.detach())
            total = total + loss / n_total
        total.backward()
        return num_gaussians, reg, local
    num_gaussians, reg, local = retry_on_capacity(attempt)()
    params = model.parameters()
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    if (world_size > 1) if all_reduce is None else all_reduce:
        flat_all_reduce([p.grad for p in params])
    return {"loss_local_views": torch.stack(local).mean() if local else torch.zeros(()), "regularization": reg.detach(),
            "#gaussians": torch.tensor(num_gaussians), "exposure": model.exposure_params.detach().exp().mean()}


def _main() -> None:
    """Launched as `python -m torch.distributed.run --nproc-per-node N -m geosplatting_amd.stage1 [iters]`: a short
    synthetic stage-1 run (ellipsoid target rendered through the same path), loss printed by rank 0."""
    import sys

    from . import synthetic as syn
    from .mesh import mesh_to_splats, vertex_normals
    from .shading import RenderableAttrs
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)
    iters = next((int(a) for a in sys.argv[1:] if a.isdigit()), 20)      # [iters] [--autograd]
    R, HW, n_views = 64, 256, max(8, world)
    cams = syn.blender_cameras(n_views, HW, HW)
    grid = FlexiCubes.from_resolution(R, device=dev, random_sdf=False, scale=1.05)
    with torch.no_grad():
        sdf_gt = (grid.vertices * torch.tensor([1.0, 1.25, 0.85], device=dev)).norm(dim=-1, keepdim=True) - 0.6
        (vg, fg), _ = grid.replace(sdf_values=sdf_gt).dual_marching_cubes()
        sp, n = mesh_to_splats(vg, fg, vertex_normals(vg, fg))
        N = sp.means.shape[0]
        attrs = RenderableAttrs(kd=torch.tensor([0.8, 0.3, 0.2], device=dev).expand(N, 3).contiguous(),
                                ks=torch.tensor([0.4, 0.1], device=dev).expand(N, 2).contiguous(), normals=n)
        env = as_splitsum(syn.make_cubemap(128).to(dev))
        gts = [attrs.splat(sp, [c], exposure=torch.tensor(1.0, device=dev), envmap=env, min_roughness=0.1,
                           max_metallic=1.0).reshape(HW, HW, 4) for c in cams]
    model = Stage1Model(R, light_resolution=128, device=dev, log2_hashmap_size=16,
                        sdf_init=grid.vertices.norm(dim=-1, keepdim=True) - 0.5)
    model.sdf_weight = 0.1
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    resync = 200
    if world > 1:
        broadcast_parameters(model)
    import time
    t0 = None
    for it in range(iters):
        if it == 2:                                          # after the table builds / allocator warm-up
            torch.cuda.synchronize(); t0 = time.time()
        m = (train_step if "--autograd" in sys.argv else train_step_fused)(
            model, cams, gts, gt_is_srgb=False, rank=rank, world_size=world)
        opt.step()
        if world > 1 and resync > 0 and (it + 1) % resync == 0:
            broadcast_parameters(model)
        if rank == 0 and (it % 5 == 0 or it == iters - 1):
            print(f"iter {it:3d}  loss(local views) {float(m['loss_local_views']):.4f}  reg {float(m['regularization']):.4f}  "
                  f"#gaussians {int(m['#gaussians'])}", flush=True)
    torch.cuda.synchronize()
    if rank == 0 and t0 is not None and iters > 2:
        print(f"{(time.time() - t0) / (iters - 2) * 1e3:.1f} ms per iteration ({n_views} views of {HW}x{HW}, grid {R}^3, {world} rank(s))")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    _main()
