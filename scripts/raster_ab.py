"""A/B timing of the compositor variants at the bench workload (one view, kernels alone, hipEvents on the launch stream).
usage: GEOSPLAT_RASTER_LANES={0,1} [GEOSPLAT_RASTER_BLOCKS=n] python scripts/raster_ab.py [level]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd._lib as L
if os.environ.get("GEOSPLAT_LIB"):
    L.LIB_PATH = os.environ["GEOSPLAT_LIB"]
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda:0")
sc = syn.sphere_scene(level, seed=1, cubemap_res=64)
cam = syn.blender_cameras(8)[0]
sp = sc.splats.to(dev)
colors = torch.rand(sp.num, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).requires_grad_(True)
args = (sp.means, sp.quats, sp.scales.exp(), torch.sigmoid(sp.opacities).squeeze(-1), colors,
        cam.view_matrix.to(dev)[None], cam.intrinsic_matrix.to(dev)[None], 800, 800)
g = torch.Generator(device=dev).manual_seed(0)
vr = torch.rand(1, 800, 800, 3, device=dev, generator=g) * 2 - 1
va = torch.rand(1, 800, 800, 1, device=dev, generator=g) * 2 - 1
from torch.profiler import profile, ProfilerActivity
for rep in range(3):
    colors.grad = None
    r, a, meta = gs.rasterization(*args)
    (r * vr).sum().add((a * va).sum()).backward()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for rep in range(10):
        colors.grad = None
        r, a, meta = gs.rasterization(*args)
        (r * vr).sum().add((a * va).sum()).backward()
    torch.cuda.synchronize()
rows = [(e.key, e.device_time_total / e.count, e.count) for e in prof.key_averages() if "raster" in e.key or "radix" in e.key.lower() or "build_stream" in e.key or "isect" in e.key or "project" in e.key or "onesweep" in e.key.lower()]
print(f"LIB={os.path.basename(L.LIB_PATH)} LANES={os.environ.get('GEOSPLAT_RASTER_LANES', '1')} BLOCKS={os.environ.get('GEOSPLAT_RASTER_BLOCKS', '4')} I={meta['flatten_ids'].numel()}")
for k, t, c in sorted(rows, key=lambda x: -x[1]):
    print(f"  {t:9.1f} us  x{c:3d}  {k[:100]}")
print("  checksum", float(r.detach().double().sum()), float(colors.grad.double().abs().sum()))
