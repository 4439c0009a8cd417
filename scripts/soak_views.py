"""Soak with a different random subset of cameras (and count) in every step, images kept: capacity protocol under changing
intersection counts, allocator growth, throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import RenderStep, params_from_scene
dev = torch.device("cuda:0")
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=40, width=800, height=800)
step = RenderStep(params_from_scene(scene, dev), prefilter=True)
up = (torch.rand(800, 800, 4) * 2 - 1).to(dev)
g = torch.Generator().manual_seed(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
views = 0; repeats = 0; t0 = time.perf_counter()
for it in range(n):
    k = 4 + int(torch.randint(0, 9, (1,), generator=g))                 # 4..12 views
    sel = [cams[int(i)] for i in torch.randperm(40, generator=g)[:k]]
    grads, images = step(sel, lambda i, img: up, all_reduce=False, keep_images=True)
    views += k
    if not step.poll_capacity():                                          # non-blocking; an overflow is reported one step later
        repeats += 1
    if (it + 1) % 100 == 0:
        torch.cuda.synchronize()
        print(f"steps {it - 99:4d}-{it + 1:4d}: {views / (time.perf_counter() - t0):7.1f} views/s, overflow reports {repeats}, truncated {step.truncated_steps}, "
              f"I_cap {step._i_cap}, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB, finite {bool(torch.isfinite(grads['means']).all())}", flush=True)
        views = 0; t0 = time.perf_counter()
