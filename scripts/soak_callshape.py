"""Soak of the call-shaped step (RenderableAttrs.splat loop + one backward): throughput and allocator state every 500 steps.
usage: python scripts/soak_callshape.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import params_from_scene
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=8, width=800, height=800)
params = params_from_scene(scene, dev)
g = torch.Generator().manual_seed(100)
ups = [(torch.rand(800, 800, 4, generator=g) * 2 - 1).to(dev) for _ in range(8)]
leaf = lambda t: t.detach().clone().requires_grad_(True)
class G: pass
gsn = G(); gsn.means, gsn.scales, gsn.quats, gsn.opacities = leaf(params.means), leaf(params.scales), leaf(params.quats), leaf(params.opacities)
attrs = gs.RenderableAttrs(kd=leaf(params.kd), ks=leaf(params.ks), normals=leaf(params.normals))
cubemap, exposure = leaf(params.cubemap), leaf(params.exposure)
leaves = [gsn.means, gsn.scales, gsn.quats, gsn.opacities, attrs.kd, attrs.ks, attrs.normals, cubemap, exposure]
opt = torch.optim.SGD(leaves, lr=1e-9)                     # a real optimizer between the steps: new parameter versions every step
def step():
    opt.zero_grad(set_to_none=True)
    env = gs.as_splitsum(cubemap)
    images = [attrs.splat(gsn, [cam], exposure=exposure, envmap=env, min_roughness=0.1, max_metallic=1.0) for cam in cams]
    loss = images[0].new_zeros(())
    for img, w in zip(images, ups):
        loss = loss + torch.dot(img.reshape(-1), w.reshape(-1))
    loss.backward()
    opt.step()
for _ in range(5):
    step()
torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter(); last = 0
for it in range(1, steps + 1):
    step()
    if it % 500 == 0 or it == steps:
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"steps {last:5d}-{it:5d}: {8 * (it - last) / dt:7.1f} views/s   allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB  reserved "
              f"{torch.cuda.memory_reserved() / 2**30:.2f} GiB  max allocated {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB  "
              f"finite {bool(torch.isfinite(gsn.means.grad).all())}", flush=True)
        t0 = time.perf_counter(); last = it
