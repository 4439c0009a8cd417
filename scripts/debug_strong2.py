"""Two ranks, one view each, views segment in a HIP graph: where does a step spend its time?  (GEOSPLAT_DEBUG_SHARE_GPU=1 + gloo on one GPU)
usage: GEOSPLAT_DEBUG_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/debug_strong2.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
import geosplatting_amd.engine as E
from geosplatting_amd.engine import RenderStep, params_from_scene
from geosplatting_amd.parallel import init_distributed_from_env
rank, world, dev = init_distributed_from_env("cuda")
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams_all = syn.blender_cameras(num=world, width=800, height=800)
cams = [cams_all[rank]]
step = RenderStep(params_from_scene(scene, dev), prefilter=True)
up = (torch.rand(800, 800, 4) * 2 - 1).to(dev)
for _ in range(2):
    step(cams, lambda i, img: up, all_reduce=(world > 1))
torch.cuda.synchronize()
assert step.poll_capacity(wait=True)
T = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
PATCH = os.environ.get("DEBUG_PATCH", "1") == "1"
if PATCH:
    E.as_splitsum_sharded = timed("prefilter fwd (sharded)", E.as_splitsum_sharded)
    E.as_splitsum_backward_sharded = timed("prefilter bwd (sharded)", E.as_splitsum_backward_sharded)
    _ar = dist.all_reduce
    dist.all_reduce = timed("all_reduce calls", _ar)
g = step.capture_views(cams, lambda i, img: up, all_reduce=(world > 1))
if PATCH:
    g.graph.replay = timed("graph replay", g.graph.replay)
for k in list(T): T[k] = 0.0
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    g()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
if rank == 0:
    print(f"per step {dt * 1e3:.1f} ms; " + "; ".join(f"{k} {v / 3 * 1e3:.1f} ms" for k, v in T.items()))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
