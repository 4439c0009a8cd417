"""Sanity run on a scene the bench does not cover: large overlapping splats (3DGS-like), 200 k Gaussians, 800x800."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = 200_000
sp = syn.random_splats(20000, seed=3)
means = (torch.rand(n, 3, generator=g) * 2 - 1).to(dev).requires_grad_(True)
quats = torch.randn(n, 4, generator=g).to(dev).requires_grad_(True)
scales = (torch.rand(n, 3, generator=g) * 0.05 + 0.005).to(dev).requires_grad_(True)      # 10-50 px footprints
opac = (torch.rand(n, generator=g) * 0.6 + 0.05).to(dev).requires_grad_(True)
colors = torch.rand(n, 3, generator=g).to(dev).requires_grad_(True)
cam = syn.blender_cameras(8)[0]
vm = cam.view_matrix.to(dev)[None]; K = cam.intrinsic_matrix.to(dev)[None]
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(); r, a, meta = gs.rasterization(means, quats, scales, opac, colors, vm, K, 800, 800); e1.record()
    (r.sum() + a.sum()).backward(); e2.record(); torch.cuda.synchronize()
    print(f"big splats: V={meta['radii'].shape[0]} I={meta['flatten_ids'].shape[0]} fwd {e0.elapsed_time(e1):.2f} ms bwd {e1.elapsed_time(e2):.2f} ms "
          f"alpha mean {a.mean().item():.3f} finite {bool(torch.isfinite(means.grad).all())}")
