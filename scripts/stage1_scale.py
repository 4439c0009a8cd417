"""One stage-1 iteration (BASELINE config 5) at full scale: a FlexiCubes grid fine enough for ~2 M Gaussians, 8 views of
800x800, fused engine step.  Prints ms per iteration; run under rocprofv3 --kernel-trace --stats for the breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd.synthetic as syn
from geosplatting_amd.stage1 import Stage1Model, train_step_fused
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 208
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
HW, n_views = 800, 8
cams = syn.blender_cameras(n_views, HW, HW)
gts = [torch.rand(HW, HW, 4, device=dev) for _ in range(n_views)]
model = Stage1Model(R, scale=1.05, light_resolution=512, device=dev, log2_hashmap_size=18)
with torch.no_grad():
    model.sdf_params.copy_(model.grid.vertices.norm(dim=-1, keepdim=True) - 0.8)
model.sdf_weight = 0.1; model.kd_regualr_perturb_std = model.ks_regualr_perturb_std = 0.01; model.kd_grad_weight = model.ks_grad_weight = 0.05
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
for it in range(iters):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.time()
    m = train_step_fused(model, cams, gts, gt_is_srgb=False)
    opt.step()
torch.cuda.synchronize()
print(f"grid {R}^3, {int(m['#gaussians'])} Gaussians, {n_views} views of {HW}x{HW}: {(time.time() - t0) / (iters - 2) * 1e3:.1f} ms per iteration")
