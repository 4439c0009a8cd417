"""CPU restatement of the loss side -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; geosplatting_amd never does.

What is restated, and from where:
  * the per-view training loss of rfstudio/trainer/geosplat_trainer.py:171-180
    (random-background blend, SSIML1Loss with ssim_lambda 0.2, 5 x mask MSE) and the sRGB PSNR metric (:191-195);
  * SSIML1Loss / SSIMLoss / L1Loss: rfstudio/loss/photometric_loss.py:73-112, rfstudio/loss/base_loss.py:26-30;
  * rgb2srgb / srgb2rgb / blend: rfstudio/graphics/_images.py:191-241,287-311
    -- PINNED by tests/golden/ref_loss.npz (outputs of the reference's own classes, scripts/make_golden_loss.py);
  * SSIM itself lives in a third-party dependency that is NOT in /root/reference and not installed here:
    torchmetrics (pyproject pin ``torchmetrics~=1.3.1``), functional.image.structural_similarity_index_measure
    with its defaults.  Its published algorithm is restated literally in `ssim_torchmetrics` below (Gaussian
    window built as exp(-(d/sigma)^2/2)/sum with kernel size int(3.5 sigma + 0.5)*2+1 = 11, reflect padding of
    (k-1)/2, grouped conv2d of the five stacked maps, variances clamped at 0 (the 1.3.2 fix, the newest
    release the pin admits), crop of the padded border, mean).  PARITY UNPINNED for this one function: no golden
    vector from torchmetrics can be produced in this image; it is cross-checked against an independent
    scipy.ndimage formulation in tests/test_oracle_cpu.py.
Works in any dtype (float64 for gradient checks through autograd).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F


def srgb2rgb(c: torch.Tensor) -> torch.Tensor:
    """graphics/_images.py:287-311"""
    return torch.where(c <= 0.04045, c / 12.92, torch.pow((c.clamp_min(0.04045) + 0.055) / 1.055, 2.4))


def rgb2srgb(c: torch.Tensor) -> torch.Tensor:
    """graphics/_images.py:217-241"""
    return torch.where(c <= 0.0031308, c * 12.92, torch.clamp(c, min=0.0031308).pow(1.0 / 2.4) * 1.055 - 0.055)


def gaussian_window(kernel_size: int, sigma: float, dtype) -> torch.Tensor:
    dist = torch.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1, dtype=dtype)
    g = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    return g / g.sum()


def ssim_torchmetrics(preds: torch.Tensor, target: torch.Tensor, sigma: float = 1.5, data_range: float = 1.0,
                      k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """preds/target [B,C,H,W] -> mean SSIM per batch element [B] (torchmetrics 1.3.x `_ssim_update` + mean)."""
    c1 = (k1 * data_range) ** 2
    c2 = (k2 * data_range) ** 2
    C = preds.shape[1]
    ks = int(3.5 * sigma + 0.5) * 2 + 1
    pad = (ks - 1) // 2
    preds = F.pad(preds, (pad, pad, pad, pad), mode="reflect")
    target = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    g = gaussian_window(ks, sigma, preds.dtype)
    kernel = (g[:, None] @ g[None, :]).expand(C, 1, ks, ks)
    stack = torch.cat((preds, target, preds * preds, target * target, preds * target))
    out = F.conv2d(stack, kernel, groups=C).split(preds.shape[0])
    mu_p2, mu_t2, mu_pt = out[0].pow(2), out[1].pow(2), out[0] * out[1]
    s_p = torch.clamp(out[2] - mu_p2, min=0.0)
    s_t = torch.clamp(out[3] - mu_t2, min=0.0)
    s_pt = out[4] - mu_pt
    full = ((2 * mu_pt + c1) * (2 * s_pt + c2)) / ((mu_p2 + mu_t2 + c1) * (s_p + s_t + c2))
    crop = full[..., pad:-pad, pad:-pad]
    return crop.reshape(crop.shape[0], -1).mean(-1)


def ssim_l1_loss(output: torch.Tensor, gt_output: torch.Tensor, ssim_lambda: float = 0.2) -> Tuple[torch.Tensor, ...]:
    """SSIML1Loss._impl (photometric_loss.py:100-112) on [H,W,3] images -> (loss, ssim_loss, l1)"""
    ssim_loss = 1 - ssim_torchmetrics(gt_output.permute(2, 0, 1)[None], output.permute(2, 0, 1)[None])[0]
    l1 = (output - gt_output).abs().mean()
    return ssim_loss * ssim_lambda + l1 * (1.0 - ssim_lambda), ssim_loss, l1


def view_loss(rgb: torch.Tensor, alpha: torch.Tensor, gt_rgba: torch.Tensor, train_bg: torch.Tensor,
              gt_is_srgb: bool = True, ssim_lambda: float = 0.2, mask_weight: float = 5.0,
              metric_bg: Optional[torch.Tensor] = None):
    """geosplat_trainer.py:171-180 for ONE view.  rgb [H,W,3] linear (the path's tone-mapped output), alpha [H,W,1],
    gt_rgba [H,W,4] (sRGB colours when gt_is_srgb), train_bg [H,W,3].
    Returns dict(loss, ssim_loss, l1, mask_mse, mse_srgb, psnr)."""
    gt_lin = srgb2rgb(gt_rgba[..., :3]) if gt_is_srgb else gt_rgba[..., :3]
    mask = gt_rgba[..., 3:]
    img1 = rgb + (1 - alpha) * train_bg
    img2 = gt_lin * mask + (1 - mask) * train_bg
    loss, ssim_loss, l1 = ssim_l1_loss(img1, img2, ssim_lambda)
    mask_mse = (mask - alpha).square().mean()
    out = {"loss": loss + mask_weight * mask_mse, "ssim_loss": ssim_loss, "l1": l1, "mask_mse": mask_mse}
    if metric_bg is not None:
        # :191-195: rgba = pbra.rgb2srgb(); rgb = rgb_s + (1 - a) * bg; PSNR(gt_rgba.blend(bg), rgb.clamp(0, 1))
        gt_s = gt_rgba[..., :3] if gt_is_srgb else rgb2srgb(gt_rgba[..., :3])
        o = (rgb2srgb(rgb.detach()) + (1 - alpha.detach()) * metric_bg).clamp(0, 1)
        t = mask * gt_s + metric_bg * (1 - mask)
        mse = (o - t).square().mean()
        out["mse_srgb"] = mse
        out["psnr"] = -10 * torch.log10(mse)
    return out
