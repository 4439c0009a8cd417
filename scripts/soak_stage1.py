"""Soak of the stage-1 loop (changing topology -> changing Gaussian counts): iteration time and allocator growth."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd.synthetic as syn
from geosplatting_amd.stage1 import Stage1Model, train_step_fused
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 80
HW, n_views = 800, 8
cams = syn.blender_cameras(n_views, HW, HW)
gts = [torch.rand(HW, HW, 4, device=dev) for _ in range(n_views)]
model = Stage1Model(R, scale=1.05, light_resolution=512, device=dev, log2_hashmap_size=18)
with torch.no_grad():
    model.sdf_params.copy_(model.grid.vertices.norm(dim=-1, keepdim=True) - 0.8)
model.sdf_weight = 0.1; model.kd_regualr_perturb_std = model.ks_regualr_perturb_std = 0.01; model.kd_grad_weight = model.ks_grad_weight = 0.05
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
t0 = time.time()
for it in range(iters):
    m = train_step_fused(model, cams, gts, gt_is_srgb=False)
    opt.step()
    if (it + 1) % 20 == 0:
        torch.cuda.synchronize()
        print(f"it {it + 1:4d}: {(time.time() - t0) / 20 * 1e3:7.1f} ms per iteration, {int(m['#gaussians'])} Gaussians, reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB, "
              f"max allocated {torch.cuda.max_memory_allocated() / 2**30:5.2f} GiB", flush=True)
        t0 = time.time()
