"""Replay time of the views-segment graph with and without a device synchronise in front of the replay (single rank)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import RenderStep, params_from_scene
dev = torch.device("cuda:0")
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 1
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=nv, width=800, height=800)
step = RenderStep(params_from_scene(scene, dev), prefilter=True)
ups = [(torch.rand(800, 800, 4) * 2 - 1).to(dev) for _ in range(nv)]
for _ in range(3):
    step(cams, lambda i, img: ups[i], all_reduce=False)
torch.cuda.synchronize()
assert step.poll_capacity(wait=True)
g = step.capture_views(cams, lambda i, img: ups[i], all_reduce=False)
def t(fn, n=1):
    t0 = time.perf_counter(); fn(); return (time.perf_counter() - t0) * 1e3 / n
sync = torch.cuda.synchronize
for i in range(3):
    g()
sync()
print("3 steps back to back, then sync: %.2f ms per step" % t(lambda: ([g() for _ in range(3)], sync()), 3))
for i in range(3):
    sync(); a = t(lambda: g.graph.replay()); b = t(sync)
    print("sync; replay() returned after %.2f ms; sync after %.2f ms" % (a, b))
for i in range(3):
    sync(); a = t(lambda: g()); b = t(sync)
    print("sync; step() returned after %.2f ms; sync after %.2f ms" % (a, b))
print("check:", g.check(), "truncated", step.truncated_steps)
# ---- the parts of step() one by one, each started on an idle device
cl = dict(zip(g.__code__.co_freevars, [c.cell_contents for c in g.__closure__]))
filter_env, slots, graph, ctx = cl["filter_env"], cl["slots"], cl["graph"], cl["ctx"]
for rep in range(2):
    sync(); a = t(lambda: filter_env()); b = t(sync); print("filter_env: returned %.2f ms, sync %.2f ms" % (a, b))
    env = filter_env(); sync()
    a = t(lambda: torch._foreach_copy_(slots, [env.base] + list(env.levels))); b = t(sync); print("copy into slots: %.2f / %.2f ms" % (a, b))
    a = t(lambda: graph.replay()); b = t(sync); print("replay: %.2f / %.2f ms" % (a, b))
    ctx["main"] = torch.cuda.current_stream(dev)
    a = t(lambda: step._finish(ctx)); b = t(sync); print("_finish: %.2f / %.2f ms" % (a, b))
    # pairs
    sync(); a = t(lambda: (graph.replay(), step._finish(ctx))); b = t(sync); print("replay + _finish: %.2f / %.2f ms" % (a, b))
    sync(); env = filter_env(); torch._foreach_copy_(slots, [env.base] + list(env.levels)); a = t(lambda: graph.replay()); b = t(sync); print("filter + copy + replay: sync %.2f ms" % b)
