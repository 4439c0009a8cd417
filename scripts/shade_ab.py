"""A/B timing of the shading backward at the bench workload (one view): GEOSPLAT_SHADE_BWD_BLOCK / GEOSPLAT_SHADE_LDS_MAXRES."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd._lib as L
if os.environ.get("GEOSPLAT_LIB"):
    L.LIB_PATH = os.environ["GEOSPLAT_LIB"]
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
dev = torch.device("cuda:0")
sc = syn.sphere_scene(7, seed=1, cubemap_res=512)
cam = syn.blender_cameras(8)[0]
with torch.no_grad():
    env = gs.as_splitsum(sc.cubemap.to(dev))
d = lambda t: t.to(dev).requires_grad_(True)
means, normals, kd, ks = d(sc.splats.means), d(sc.normals), d(sc.kd), d(sc.ks)
envl = gs.TextureSplitSum(env.base.requires_grad_(True), [l.requires_grad_(True) for l in env.levels])
col = gs.shade(means, normals, kd, ks, cam.c2w[:, 3].to(dev).contiguous(), envl, min_roughness=0.1, max_metallic=1.0)
# realistic cotangent: zero for the Gaussians that reach no pixel (about half), from one rasterizer backward
sp = sc.splats.to(dev)
c2 = col.detach().clone().requires_grad_(True)
r, a, _ = gs.rasterization(sp.means, sp.quats, sp.scales.exp(), torch.sigmoid(sp.opacities).squeeze(-1), c2, cam.view_matrix.to(dev)[None],
                           cam.intrinsic_matrix.to(dev)[None], 800, 800)
(r * torch.rand_like(r)).sum().backward()
v = c2.grad.clone()
print("non-zero colour gradients:", int((v.abs().sum(-1) > 0).sum()), "of", v.shape[0])
from torch.profiler import profile, ProfilerActivity
for rep in range(3):
    col.backward(v, retain_graph=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for rep in range(10):
        col.backward(v, retain_graph=True)
    torch.cuda.synchronize()
print(f"BLOCK={os.environ.get('GEOSPLAT_SHADE_BWD_BLOCK', '256')} LDS_MAXRES={os.environ.get('GEOSPLAT_SHADE_LDS_MAXRES', '16')} PRIV={os.environ.get('GEOSPLAT_SHADE_PRIV', '0')}")
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
    if "shade" in e.key or "priv_reduce" in e.key:
        print(f"  {e.device_time_total / e.count:9.1f} us  x{e.count:3d}  {e.key[:90]}")
print("  checksum", float(sum(l.grad.double().abs().sum() for l in envl.levels)), float(kd.grad.double().abs().sum()))
