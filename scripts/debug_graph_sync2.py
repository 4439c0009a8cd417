import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import RenderStep, params_from_scene
dev = torch.device("cuda:0")
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=1, width=800, height=800)
step = RenderStep(params_from_scene(scene, dev), prefilter=True)
ups = [(torch.rand(800, 800, 4) * 2 - 1).to(dev)]
for _ in range(3):
    step(cams, lambda i, img: ups[i], all_reduce=False)
torch.cuda.synchronize()
assert step.poll_capacity(wait=True)
g = step.capture_views(cams, lambda i, img: ups[i], all_reduce=False)
sync = torch.cuda.synchronize
for i in range(3):
    g()
sync()
t0 = time.perf_counter(); g(); sync(); print("step after sync: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); g(); sync(); print("step after sync: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
