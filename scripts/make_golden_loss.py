#!/usr/bin/env python3
"""Generate tests/golden/ref_loss.npz from the reference's OWN loss glue (build container only):

    cd /tmp && PYTHONPATH=/tmp/stubs:/root/reference python /root/repo/scripts/make_golden_loss.py

What it pins: RGBAImages.srgb2rgb / PBRAImages.rgb2srgb / RGBAImages.blend (rfstudio/graphics/_images.py) and
SSIML1Loss / PSNRLoss / L1Loss (rfstudio/loss/photometric_loss.py, base_loss.py) composed exactly as
rfstudio/trainer/geosplat_trainer.py:171-195 composes them for one view.  torchmetrics is absent from this
image, so the reference's `structural_similarity_index_measure` import is served by this repo's restatement
(oracle/loss_ref.py::ssim_torchmetrics) -- the fixture pins the glue around SSIM, not SSIM itself.
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
from oracle import loss_ref                                              # noqa: E402

for name in ["open3d", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
             "cv2", "pyexr", "trimesh", "nvdiffrast", "nvdiffrast.torch", "kornia", "kornia.filters", "gsplat",
             "torchmetrics", "torchmetrics.functional", "torchmetrics.functional.image", "ffmpegcv", "nerfacc", "tyro",
             "skimage", "skimage.measure", "rfviser", "viser", "appdirs", "huggingface_hub", "pytorch3d",
             "pytorch3d.loss", "pytorch3d.structures", "rfstudio.graphics._mesh._optix", "rfstudio.graphics._mesh._splitsum",
             "tinycudann", "plotext", "imageio", "lpips", "matplotlib", "matplotlib.pyplot", "viser.transforms",
             "rfviser.transforms", "torchmetrics.image", "torchmetrics.image.lpip"]:
    sys.modules.setdefault(name, MagicMock())
sys.modules["torchmetrics.functional.image"].structural_similarity_index_measure = \
    lambda preds, target, data_range=1.0: loss_ref.ssim_torchmetrics(preds, target, data_range=data_range)[0]

from rfstudio.graphics import PBRAImages, RGBAImages, RGBImages           # noqa: E402
from rfstudio.loss import PSNRLoss, SSIML1Loss                            # noqa: E402

g = torch.Generator().manual_seed(77)
H, W = 40, 56
# a rendered view (linear rgb premultiplied by alpha, soft silhouette) and an sRGB ground truth with a mask
yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
alpha = torch.sigmoid((0.6 - (xx * xx + yy * yy).sqrt()) * 12)[..., None]
rgb = (0.5 + 0.5 * torch.sin(torch.stack([3 * xx, 4 * yy, 5 * (xx + yy)], -1))) * alpha * 0.9
rgb = rgb + 0.02 * torch.rand(H, W, 3, generator=g) * alpha
mask = (torch.sigmoid((0.55 - (xx * xx + 1.2 * yy * yy).sqrt()) * 30) > 0.5).float()[..., None]
gt_srgb = torch.rand(H, W, 3, generator=g) * 0.3 + 0.35 + 0.3 * torch.cos(torch.stack([2 * xx, 3 * yy, xx - yy], -1))
gt_srgb = gt_srgb.clamp(0, 1)
gt_srgb[:4, :4] = 0.02                                                   # below the 0.04045 knee
gt_rgba = torch.cat([gt_srgb, mask], -1)
train_bg = torch.rand(H, W, 3, generator=g)
bg_color = torch.tensor([1.0, 1.0, 1.0])

pbra = PBRAImages([torch.cat([rgb, alpha], -1)])
gt = RGBAImages([gt_rgba])
rgba_srgb = pbra.rgb2srgb()
gt_lin = gt.srgb2rgb()
# geosplat_trainer.py:171-180, one view, with the random background fixed to `train_bg`
pb, gp = next(iter(zip(pbra, gt_lin, strict=True)))
m = gp[..., 3:]
img1 = pb[..., :3] + (1 - pb[..., 3:]) * train_bg
img2 = gp[..., :3] * m + (1 - m) * train_bg
ssim_l1 = SSIML1Loss()._impl(img1, img2)
loss = ssim_l1 + 5 * (m - pb[..., 3:]).square().mean()
# :191-195
rgb_metric = RGBImages([it[..., :3] + (1 - it[..., 3:]) * bg_color for it in rgba_srgb.detach()])
psnr = PSNRLoss()(gt.blend(bg_color), rgb_metric.clamp(0, 1))

np.savez_compressed(os.path.join(OUT, "ref_loss.npz"), rgb=rgb.numpy(), alpha=alpha.numpy(), gt_rgba=gt_rgba.numpy(),
                    train_bg=train_bg.numpy(), bg_color=bg_color.numpy(), rgb2srgb=rgba_srgb.get(0).numpy(),
                    srgb2rgb=gt_lin.get(0).numpy(), gt_blend=gt.blend(bg_color).get(0).numpy(), img1=img1.numpy(),
                    img2=img2.numpy(), ssim_l1=float(ssim_l1), loss=float(loss), psnr=float(psnr))
print("loss", float(loss), "ssim_l1", float(ssim_l1), "psnr", float(psnr))
