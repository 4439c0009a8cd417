"""The bench step through the reference's call shape (as_splitsum -> loop of RenderableAttrs.splat -> one backward), alone:
host time per phase against GPU time per step; the command rocprofv3 traces for profiles/r05_callshape_*.
usage: python scripts/callshape_step.py [steps] [views]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import params_from_scene
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_views = int(sys.argv[2]) if len(sys.argv) > 2 else 8
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=n_views, width=800, height=800)
params = params_from_scene(scene, dev)
g = torch.Generator().manual_seed(100)
ups = [(torch.rand(800, 800, 4, generator=g) * 2 - 1).to(dev) for _ in range(n_views)]
leaf = lambda t: t.detach().clone().requires_grad_(True)
class G: pass
gsn = G(); gsn.means, gsn.scales, gsn.quats, gsn.opacities = leaf(params.means), leaf(params.scales), leaf(params.quats), leaf(params.opacities)
attrs = gs.RenderableAttrs(kd=leaf(params.kd), ks=leaf(params.ks), normals=leaf(params.normals))
cubemap, exposure = leaf(params.cubemap), leaf(params.exposure)
leaves = [gsn.means, gsn.scales, gsn.quats, gsn.opacities, attrs.kd, attrs.ks, attrs.normals, cubemap, exposure]
T = {"prefilter": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0}
def step(clock=False):
    a = time.perf_counter()
    for t in leaves:
        t.grad = None
    env = gs.as_splitsum(cubemap)
    b = time.perf_counter()
    images = [attrs.splat(gsn, [cam], exposure=exposure, envmap=env, min_roughness=0.1, max_metallic=1.0) for cam in cams]
    c = time.perf_counter()
    loss = images[0].new_zeros(())
    for img, w in zip(images, ups):
        loss = loss + torch.dot(img.reshape(-1), w.reshape(-1))
    d = time.perf_counter()
    loss.backward()
    e = time.perf_counter()
    if clock:
        T["prefilter"] += b - a; T["forward"] += c - b; T["loss"] += d - c; T["backward"] += e - d
for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(True)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"call-shaped step: {1e3 * t_all / steps:.2f} ms per step = {n_views * steps / t_all:.1f} views/s; host enqueue {1e3 * t_enq / steps:.2f} ms per step: "
      + ", ".join(f"{k} {1e3 * v / steps:.2f}" for k, v in T.items()))
if not os.environ.get("CALLSHAPE_NO_SYNC_LOOP"):
    ts = []
    for _ in range(steps):
        a = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
    print(f"step by step with a synchronise after each: mean {1e3 * sum(ts) / steps:.2f} ms, min {1e3 * min(ts):.2f}")
if os.environ.get("CALLSHAPE_HOST_TIMES"):
    # host clock at the entry of every splat() of ONE step that starts on an idle GPU, and the GPU's clock (events) at the end of each
    # view's forward on the caller's stream: how far ahead of the GPU does the host enqueue the front chains?
    torch.cuda.synchronize()
    for t in leaves:
        t.grad = None
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    h0 = time.perf_counter()
    env = gs.as_splitsum(cubemap)
    hs, evs, images = [], [], []
    for cam in cams:
        hs.append(time.perf_counter() - h0)
        images.append(attrs.splat(gsn, [cam], exposure=exposure, envmap=env, min_roughness=0.1, max_metallic=1.0))
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    h_end = time.perf_counter() - h0
    torch.cuda.synchronize()
    print("host ms at splat() entry:", " ".join(f"{1e3 * x:.2f}" for x in hs), f"| all enqueued at {1e3 * h_end:.2f}")
    print("GPU  ms at forward end  :", " ".join(f"{e0.elapsed_time(e):.2f}" for e in evs))
if os.environ.get("CALLSHAPE_CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    print(torch.cuda.memory_summary(abbreviated=True))
if os.environ.get("CALLSHAPE_HOSTTRACE"):
    from geosplatting_amd import viewbatch as vb
    log = []
    def wrap(obj, name):
        f = getattr(obj, name)
        def g(*a, **k):
            t = time.perf_counter(); r = f(*a, **k); log.append((name, t, time.perf_counter())); return r
        setattr(obj, name, g)
    wrap(vb, "_view_backward"); wrap(vb, "_poll_unchecked"); wrap(vb._Step, "launch_tail"); wrap(vb._Step, "grads"); wrap(vb, "_view_forward")
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"one step: host {1e3 * (t1 - t0):.2f} ms, GPU done at {1e3 * (t2 - t0):.2f} ms")
    for n, a, b in log:
        print(f"  {n:16s} {1e3 * (a - t0):8.3f} -> {1e3 * (b - t0):8.3f}")
