import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
dev = torch.device("cuda:0")
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
res = int(sys.argv[2]) if len(sys.argv) > 2 else 160
sc = syn.sphere_scene(lvl, seed=1, cubemap_res=64)
cam = syn.blender_cameras(8, res, res)[0]
sp = sc.splats.to(dev)
colors = torch.rand(sp.num, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
args = (sp.means, sp.quats, sp.scales.exp(), torch.sigmoid(sp.opacities).squeeze(-1), colors, cam.view_matrix.to(dev)[None], cam.intrinsic_matrix.to(dev)[None], res, res)
with torch.no_grad():
    r, a, meta = gs.rasterization(*args)
torch.cuda.synchronize()
print("fwd ok: nan", int(torch.isnan(r).sum()), "sum", float(r.double().sum()), "alpha sum", float(a.double().sum()), "I", meta["flatten_ids"].numel())
