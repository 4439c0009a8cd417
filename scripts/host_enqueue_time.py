"""How long does the HOST need to enqueue one engine step (no synchronisation inside), against the GPU time of the step?
usage: python scripts/host_enqueue_time.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import RenderStep, params_from_scene
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=8, width=800, height=800)
step = RenderStep(params_from_scene(scene, dev), prefilter=True)
g = torch.Generator().manual_seed(100)
ups = [(torch.rand(800, 800, 4, generator=g) * 2 - 1).to(dev) for _ in range(8)]
for _ in range(5):
    step(cams, lambda i, img: ups[i], all_reduce=False)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter()
    step(cams, lambda i, img: ups[i], all_reduce=False)
    host.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue per step: {1e3 * sum(host) / steps:.2f} ms (min {1e3 * min(host):.2f}, max {1e3 * max(host):.2f});  all {steps} steps enqueued after "
      f"{1e3 * t_enq:.1f} ms, GPU done after {1e3 * t_all:.1f} ms ({1e3 * t_all / steps:.2f} ms per step)")
# the same with the profiler's view of python time
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    step(cams, lambda i, img: ups[i], all_reduce=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
# one step at a time (the host starts every step with empty queues): GPU time of a step when the host is far ahead of it
torch.cuda.synchronize()
ts = []
for _ in range(steps):
    a = time.perf_counter()
    step(cams, lambda i, img: ups[i], all_reduce=False)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - a)
print(f"step by step with a synchronise after each: mean {1e3 * sum(ts) / steps:.2f} ms, min {1e3 * min(ts):.2f}")
