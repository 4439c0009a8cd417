"""Experiment: where the host spends loss.backward() of the call-shaped step (cProfile over 10 steps; the autograd engine runs the
nodes of a CUDA graph-less backward on its device thread, so the python frames of the node backwards are profiled with threading.setprofile)."""
import os, sys, time, cProfile, pstats, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import params_from_scene
from geosplatting_amd import viewbatch as VB, front as F
dev = torch.device("cuda:0")
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=8, width=800, height=800)
params = params_from_scene(scene, dev)
leaf = lambda t: t.detach().clone().requires_grad_(True)
class G: pass
gsn = G(); gsn.means, gsn.scales, gsn.quats, gsn.opacities = leaf(params.means), leaf(params.scales), leaf(params.quats), leaf(params.opacities)
attrs = gs.RenderableAttrs(kd=leaf(params.kd), ks=leaf(params.ks), normals=leaf(params.normals))
cubemap, exposure = leaf(params.cubemap), leaf(params.exposure)
leaves = [gsn.means, gsn.scales, gsn.quats, gsn.opacities, attrs.kd, attrs.ks, attrs.normals, cubemap, exposure]
ups = [(torch.rand(800, 800, 4) * 2 - 1).to(dev) for _ in range(8)]
# wall-clock stamps inside the node backwards
stamps = []
orig_vb, orig_lt, orig_poll = VB._view_backward, VB._Step.launch_tail, VB._poll_unchecked
def vb(*a, **k):
    t = time.perf_counter(); r = orig_vb(*a, **k); stamps.append(("view_backward", t, time.perf_counter())); return r
def lt(self, final):
    t = time.perf_counter(); r = orig_lt(self, final); stamps.append(("launch_tail" + ("_final" if final else ""), t, time.perf_counter())); return r
def poll(*a, **k):
    t = time.perf_counter(); r = orig_poll(*a, **k); stamps.append(("poll", t, time.perf_counter())); return r
VB._view_backward, VB._Step.launch_tail, VB._poll_unchecked = vb, lt, poll
def step(report=False):
    for t in leaves:
        t.grad = None
    env = gs.as_splitsum(cubemap)
    images = [attrs.splat(gsn, [cam], exposure=exposure, envmap=env, min_roughness=0.1, max_metallic=1.0) for cam in cams]
    loss = images[0].new_zeros(())
    for img, w in zip(images, ups):
        loss = loss + torch.dot(img.reshape(-1), w.reshape(-1))
    del stamps[:]
    t0 = time.perf_counter()
    loss.backward()
    t1 = time.perf_counter()
    if report:
        print(f"backward() {1e3 * (t1 - t0):.2f} ms on the host:")
        for name, a, b in stamps:
            print(f"   {name:20s} starts {1e3 * (a - t0):7.2f}  takes {1e3 * (b - a):6.2f} ms")
for _ in range(6):
    step()
step(True)
torch.cuda.synchronize()
step(True)
