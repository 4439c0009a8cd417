import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import RenderStep, params_from_scene
dev = torch.device("cuda:0")
variant = sys.argv[1]
level = int(os.environ.get("DBG_LEVEL", "7"))
scene = syn.sphere_scene(level, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=1, width=800, height=800)
step = RenderStep(params_from_scene(scene, dev), prefilter=(variant != "whole_noprefilter"))
ups = [(torch.rand(800, 800, 4) * 2 - 1).to(dev)]
for _ in range(3):
    step(cams, lambda i, img: ups[i], all_reduce=False)
torch.cuda.synchronize()
assert step.poll_capacity(wait=True)
if variant == "views":
    g = step.capture_views(cams, lambda i, img: ups[i], all_reduce=False); replay = g.graph.replay
else:
    g = step.capture(cams, lambda i, img: ups[i]); replay = g.graph.replay if hasattr(g, "graph") else g
sync = torch.cuda.synchronize
one = torch.zeros(1024, device=dev)
def run(name, fn):
    sync(); t0 = time.perf_counter(); fn(); sync(); print("%-12s %-30s %10.2f ms" % (variant, name, (time.perf_counter() - t0) * 1e3), flush=True)
run("replay", replay); run("replay", replay)
run("eager kernel; replay", lambda: (one.add_(1), replay()))
run("replay", replay)
