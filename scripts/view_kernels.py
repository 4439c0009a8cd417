"""One view at the bench size through geosplatting_amd.rasterization (single stream, nothing overlapped), a few repetitions:
run under `rocprofv3 --kernel-trace --stats` to get every kernel's time ALONE.  python scripts/view_kernels.py [level=7] [reps=6]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import geosplatting_amd as gs
import geosplatting_amd.synthetic as syn

dev = torch.device("cuda", 0)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sc = syn.sphere_scene(level, seed=1, cubemap_res=64)
cam = syn.blender_cameras(8, 800, 800)[1]
sp = sc.splats
means = sp.means.to(dev).requires_grad_(True); quats = sp.quats.to(dev).requires_grad_(True)
scales = sp.scales.to(dev).exp().requires_grad_(True); opac = torch.sigmoid(sp.opacities.to(dev)).squeeze(-1).requires_grad_(True)
col = torch.rand(sp.num, 3, device=dev).requires_grad_(True)
vm = cam.view_matrix.to(dev)[None]; K = cam.intrinsic_matrix.to(dev)[None]
v = torch.rand(1, 800, 800, 3, device=dev)
for _ in range(reps):
    r, a, meta = gs.rasterization(means, quats, scales, opac, col, vm, K, 800, 800)
    (r * v).sum().backward()
    torch.cuda.synchronize()
print("V", meta["gaussian_ids"].numel(), "I", meta["flatten_ids"].numel())
