"""Is the host ahead of the GPU?  Enqueue time of K engine steps (no synchronisation) against their wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import RenderStep, params_from_scene
dev = torch.device("cuda:0")
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=8, width=800, height=800)
step = RenderStep(params_from_scene(scene, dev), prefilter=True)
ups = [(torch.rand(800, 800, 4) * 2 - 1).to(dev) for _ in range(8)]
for _ in range(40):
    step(cams, lambda i, img: ups[i], all_reduce=False)
torch.cuda.synchronize()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for rep in range(3):
    t0 = time.perf_counter(); marks = []
    for _ in range(K):
        step(cams, lambda i, img: ups[i], all_reduce=False)
        marks.append(time.perf_counter())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    d = [1e3 * (b - a) for a, b in zip([t0] + marks[:-1], marks)]
    print(f"enqueue of {K} steps {1e3 * (t1 - t0):.1f} ms ({1e3 * (t1 - t0) / K:.2f} per step; first three {d[0]:.2f} {d[1]:.2f} {d[2]:.2f}, last {d[-1]:.2f}); wall {1e3 * (t2 - t0):.1f} ms ({1e3 * (t2 - t0) / K:.2f} per step); GPU still busy for {1e3 * (t2 - t1):.2f} ms after the last enqueue")
