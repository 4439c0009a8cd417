"""Soak: many engine steps back to back; throughput drift and allocator growth."""
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
for _ in range(5):
    step(cams, lambda i, img: ups[i], all_reduce=False)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for chunk in range(4):
    t0 = time.perf_counter()
    for _ in range(n // 4):
        step(cams, lambda i, img: ups[i], all_reduce=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"steps {chunk * (n // 4):4d}-{(chunk + 1) * (n // 4):4d}: {8 * (n // 4) / dt:7.1f} views/s   allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB  reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB  "
          f"max allocated {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB  ok={step.poll_capacity(wait=True)} truncated={step.truncated_steps}")
