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
cl = dict(zip(g.__code__.co_freevars, [c.cell_contents for c in g.__closure__]))
filter_env, slots, graph, ctx = cl["filter_env"], cl["slots"], cl["graph"], cl["ctx"]
one = torch.zeros(1024, device=dev)
def run(name, fn):
    sync(); t0 = time.perf_counter(); fn(); sync(); print("%-44s %10.2f ms" % (name, (time.perf_counter() - t0) * 1e3), flush=True)
for i in range(3):
    g()
sync()
env = filter_env(); sync()
big = torch.zeros(64 << 20, device=dev)
gstream = torch.cuda.Stream(device=dev)
def replay_on_side():
    ev = torch.cuda.Event(); ev.record()
    gstream.wait_event(ev)
    with torch.cuda.stream(gstream):
        graph.replay()
    torch.cuda.current_stream().wait_stream(gstream)
which = sys.argv[1] if len(sys.argv) > 1 else "a"
run("E1 replay", lambda: graph.replay())
if which == "a":
    run("tiny kernel; sleep 20 ms; replay", lambda: (one.add_(1), time.sleep(0.02), graph.replay()))
    run("E1 replay", lambda: graph.replay())
    run("long kernels (5 ms); replay", lambda: ([big.add_(1) for _ in range(40)], graph.replay()))
    run("E1 replay", lambda: graph.replay())
    run("tiny kernel; replay on a side stream", lambda: (one.add_(1), replay_on_side()))
    run("E1 replay", lambda: graph.replay())
    run("tiny kernel; replay", lambda: (one.add_(1), graph.replay()))
    run("E1 replay", lambda: graph.replay())
