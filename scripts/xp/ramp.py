"""Experiment: throughput of the engine in buckets of 20 steps from a cold process (how long the ramp to the steady state takes)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import RenderStep, params_from_scene
dev = torch.device("cuda:0")
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=8, width=800, height=800)
step = RenderStep(params_from_scene(scene, dev), prefilter=True)
ups = [(torch.rand(800, 800, 4) * 2 - 1).to(dev) for _ in range(8)]
for _ in range(3):
    step(cams, lambda i, img: ups[i], all_reduce=False)
torch.cuda.synchronize()
out = []
for b in range(25):
    t0 = time.perf_counter()
    for _ in range(20):
        step(cams, lambda i, img: ups[i], all_reduce=False)
    torch.cuda.synchronize()
    out.append(160 / (time.perf_counter() - t0))
print(" ".join(f"{x:.0f}" for x in out))
# the same with a synchronize after every step (the bench's timed region has one only at its ends)
