"""D = 14 deferred rasterization (rfstudio/model/geosplat.py:276-295) forward + backward of one view at the bench size, alone:
the command rocprofv3 traces for the per-kernel split.   usage: python scripts/d14_step.py [iters] [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import params_from_scene
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
D = int(sys.argv[2]) if len(sys.argv) > 2 else 14
scene = syn.sphere_scene(7, seed=1, cubemap_res=64, device=dev)
cam = syn.blender_cameras(num=8, width=800, height=800)[0]
p = params_from_scene(scene, dev)
g = torch.Generator().manual_seed(14)
feats = torch.rand(p.means.shape[0], D, generator=g).to(dev).requires_grad_(True)
means = p.means.detach().clone().requires_grad_(True)
scales = p.scales.exp(); opac = torch.sigmoid(p.opacities).squeeze(-1)
vm, K = cam.view_matrix.to(dev)[None], cam.intrinsic_matrix.to(dev)[None]
def one():
    r, a, _ = gs.rasterization(means, p.quats, scales, opac, feats, vm, K, 800, 800)
    (r.sum() + a.sum()).backward()
    means.grad = None; feats.grad = None
one(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    one()
torch.cuda.synchronize()
print(f"D={D}: {(time.perf_counter() - t0) / iters * 1e3:.2f} ms per view fwd+bwd")
