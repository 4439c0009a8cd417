"""Loss side of the path on the GPU: what rfstudio/trainer/geosplat_trainer.py:171-195
does with each rendered view -- random-background blend, 0.2 x (1 - SSIM) + 0.8 x L1, 5 x mask MSE, and the sRGB
PSNR metric -- as ONE C-ABI call that returns the value and the gradient (csrc/gs_loss.hip).  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib


def _photo_loss_raw(rgb: Tensor, alpha: Tensor, gt_rgba: Tensor, train_bg: Tensor, metric_bg: Optional[Tensor],
                    gt_is_srgb: bool, ssim_lambda: float, mask_weight: float, grad_scale: float, want_grad: bool):
    _lib.require_cuda(rgb, alpha, gt_rgba, train_bg, metric_bg)
    H, W = rgb.shape[0], rgb.shape[1]
    if rgb.shape != (H, W, 3) or alpha.numel() != H * W or gt_rgba.shape != (H, W, 4) or train_bg.shape != (H, W, 3):
        raise _lib.GeoSplatHipError("photo_loss expects rgb[H,W,3], alpha[H,W,1], gt_rgba[H,W,4], train_bg[H,W,3]")
    rgb, alpha, gt_rgba, train_bg = (t.detach().contiguous().float() for t in (rgb, alpha, gt_rgba, train_bg))
    mb = None if metric_bg is None else metric_bg.detach().contiguous().float()
    out = torch.empty(6, device=rgb.device)
    v_rgb = torch.empty_like(rgb) if want_grad else None
    v_alpha = torch.empty_like(alpha) if want_grad else None
    nbytes = _lib.lib().gs_photo_loss_ws_bytes(W, H)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=rgb.device)
    _lib.check(_lib.lib().gs_photo_loss(W, H, _lib.ptr(rgb), _lib.ptr(alpha), _lib.ptr(gt_rgba), int(bool(gt_is_srgb)),
                                        _lib.ptr(train_bg), _lib.ptr(mb), _lib.f32(ssim_lambda), _lib.f32(mask_weight),
                                        _lib.f32(grad_scale), _lib.ptr(out), _lib.ptr(v_rgb), _lib.ptr(v_alpha),
                                        _lib.ptr(ws), C.c_size_t(nbytes), _lib.stream()), "gs_photo_loss")
    return out, v_rgb, v_alpha


class _PhotoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, alpha, gt_rgba, train_bg, metric_bg, gt_is_srgb, ssim_lambda, mask_weight):
        out, v_rgb, v_alpha = _photo_loss_raw(rgb, alpha, gt_rgba, train_bg, metric_bg, gt_is_srgb, ssim_lambda,
                                              mask_weight, 1.0, True)
        ctx.save_for_backward(v_rgb, v_alpha)
        ctx.alpha_shape = alpha.shape
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        v_rgb, v_alpha = ctx.saved_tensors
        return v_rgb * g_loss, (v_alpha * g_loss).reshape(ctx.alpha_shape), None, None, None, None, None, None


def photo_loss(rgb: Tensor, alpha: Tensor, gt_rgba: Tensor, train_bg: Optional[Tensor] = None, *,
               metric_bg: Optional[Tensor] = None, gt_is_srgb: bool = True, ssim_lambda: float = 0.2,
               use_mask_loss: bool = True) -> Tuple[Tensor, Dict[str, Tensor]]:
    """One view of geosplat_trainer.py:171-180.  rgb [H,W,3] + alpha [H,W,1] = the path's output for this view,
    gt_rgba [H,W,4] the dataset image (sRGB + mask); train_bg defaults to torch.rand_like(rgb) as the trainer draws
    it.  Returns (loss, metrics) with metrics = ssim_loss, l1, mask_mse and, when metric_bg [3] is given, the
    sRGB-space mse / psnr of :191-195."""
    if train_bg is None:
        train_bg = torch.rand_like(rgb)
    loss, out = _PhotoLoss.apply(rgb, alpha, gt_rgba, train_bg, metric_bg, gt_is_srgb, float(ssim_lambda),
                                 5.0 if use_mask_loss else 0.0)
    metrics = {"ssim_loss": out[1], "l1": out[2], "mask_mse": out[3]}
    if metric_bg is not None:
        metrics["mse_srgb"] = out[4]; metrics["psnr"] = out[5]
    return loss, metrics


def photo_loss_and_grad(rgb: Tensor, alpha: Tensor, gt_rgba: Tensor, train_bg: Tensor, *, grad_scale: float = 1.0,
                        metric_bg: Optional[Tensor] = None, gt_is_srgb: bool = True, ssim_lambda: float = 0.2,
                        use_mask_loss: bool = True) -> Tuple[Tensor, Tensor, Tensor]:
    """Fused form for a hand-scheduled step: returns (out[6], v_rgb, v_alpha) with the gradients already scaled by
    grad_scale (e.g. 1 / number of views, geosplat_trainer.py:180)."""
    return _photo_loss_raw(rgb, alpha, gt_rgba, train_bg, metric_bg, gt_is_srgb, ssim_lambda,
                           5.0 if use_mask_loss else 0.0, grad_scale, True)


class TrainerUpstream:
    """`upstream(i, image)` callback for engine.RenderStep: the trainer's per-view loss
    (geosplat_trainer.py:171-180, mean over the views of the step) evaluated on the GPU; hands back d(loss)/d(image)
    and keeps the per-view values in `.out` ([num_local_views, 6] device tensor, no host sync)."""

    def __init__(self, gt_rgba, num_views_total: int, metric_bg: Optional[Tensor] = None, gt_is_srgb: bool = True,
                 ssim_lambda: float = 0.2, use_mask_loss: bool = True, seed: int = 0, train_bg=None,
                 generator: Optional[torch.Generator] = None, device=None):
        self.gt = gt_rgba                              # sequence of [H,W,4] device tensors, local views in order
        self.scale = 1.0 / float(num_views_total)
        self.metric_bg, self.gt_is_srgb, self.ssim_lambda, self.use_mask_loss = metric_bg, gt_is_srgb, ssim_lambda, use_mask_loss
        dev = gt_rgba[0].device if len(gt_rgba) else torch.device(device if device is not None else "cuda")
        # `generator`: a persistent stream owned by the caller (fresh noise every iteration, as the trainer's rand_like)
        self.gen = generator if generator is not None else torch.Generator(device=dev).manual_seed(seed)
        self.train_bg = train_bg                       # optional fixed backgrounds (tests); default: the trainer's rand_like
        self.out = torch.zeros(len(gt_rgba), 6, device=dev)

    def __call__(self, i: int, image: Tensor) -> Tensor:
        img = image.reshape(image.shape[-3], image.shape[-2], 4)
        rgb = img[..., :3].contiguous(); alpha = img[..., 3:].contiguous()
        bg = self.train_bg[i] if self.train_bg is not None else torch.rand(rgb.shape, generator=self.gen, device=rgb.device)
        out, v_rgb, v_alpha = photo_loss_and_grad(rgb, alpha, self.gt[i], bg, grad_scale=self.scale, metric_bg=self.metric_bg,
                                                  gt_is_srgb=self.gt_is_srgb, ssim_lambda=self.ssim_lambda,
                                                  use_mask_loss=self.use_mask_loss)
        self.out[i] = out
        return torch.cat((v_rgb, v_alpha), dim=-1).reshape(image.shape)

    def mean_loss(self) -> Tensor:
        return self.out[:, 0].sum() * self.scale
