"""Driver for PMC passes: one view of shade+rasterize forward/backward at the bench workload, 3 repetitions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda:0")
sc = syn.sphere_scene(level, seed=1, cubemap_res=512)
cam = syn.blender_cameras(8)[0]
with torch.no_grad():
    env0 = gs.as_splitsum(sc.cubemap.to(dev))
d = lambda t: t.to(dev).requires_grad_(True)
sp = sc.splats
class G: pass
g = G(); g.means = d(sp.means); g.scales = d(sp.scales); g.quats = d(sp.quats); g.opacities = d(sp.opacities)
attrs = gs.RenderableAttrs(kd=d(sc.kd), ks=d(sc.ks), normals=d(sc.normals))
env = gs.TextureSplitSum(env0.base.requires_grad_(True), [l.requires_grad_(True) for l in env0.levels])
v = torch.rand(800, 800, 4, device=dev)
for rep in range(3):
    img = attrs.splat(g, [cam], exposure=torch.tensor(1.0, device=dev), envmap=env, min_roughness=0.1, max_metallic=1.0)
    img.backward(v)
torch.cuda.synchronize()
print("done")
