"""Debug: capacity-protocol stages against the exact stages on one view."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd.synthetic as syn
import importlib; R = importlib.import_module("geosplatting_amd.rasterization")
dev = torch.device("cuda:0")
sc = syn.sphere_scene(3, seed=2, cubemap_res=64)
cam = syn.blender_cameras(4, 160, 160)[0]
sp = sc.splats.to(dev)
col = torch.rand(sp.num, 3, device=dev)
args = (sp.means, sp.quats, sp.scales.exp(), torch.sigmoid(sp.opacities).squeeze(-1).contiguous(), col, cam.view_matrix.to(dev), cam.intrinsic_matrix.to(dev), 160, 160, 16, 0.3, 0.01, 1e10, 0.0)
pr = R._project_stage(*args)
st, V, I, D, whs = R._bin_stage(pr)
st = R._prepare_stage(st, V, I, D, whs)
r0, a0, s0, _, _ = R._composite_stage(st, V, I, D, whs, None)
print("exact V, I, N", V, I, sp.num)
pr = R._project_stage(*args)
status = torch.zeros(3, dtype=torch.int64, device=dev)
cap = ((int(I * 1.25) + 65535) // 65536) * 65536
st1, Vc, Ic, D, whs = R._bin_stage_cap(pr, cap, status)
torch.cuda.synchronize()
print("status", status.tolist(), "counts", st1["counts"].tolist())
print("flatten equal", torch.equal(st1["flatten_ids"][:I], s0["flatten_ids"]), "ids equal", torch.equal(st1["isect_ids"][:I], s0["isect_ids"]),
      "offsets equal", torch.equal(st1["isect_offsets"], s0["isect_offsets"]))
st1 = R._prepare_stage_cap(st1, Vc, Ic, D, whs)
r1, a1, s1 = R._composite_stage_cap(st1, Vc, Ic, D, whs, None)
torch.cuda.synchronize()
print("render diff", (r1 - r0).abs().max().item(), "alpha diff", (a1 - a0).abs().max().item(), "last_ids equal", torch.equal(s1["last_ids"], s0["last_ids"]))
