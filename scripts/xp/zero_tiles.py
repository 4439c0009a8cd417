"""Experiment: how many 'visible' Gaussians own no tile at all under the tight rectangles (candidates for dropping in the front)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd import _lib as L, front as F
from geosplatting_amd.engine import params_from_scene
from geosplatting_amd.shading import _MODE, _make_env, get_fg_lut
dev = torch.device("cuda:0")
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=8, width=800, height=800)
p = params_from_scene(scene, dev)
with torch.no_grad():
    env = gs.as_splitsum(p.cubemap)
e = _make_env(get_fg_lut(dev), gs.TextureSplitSum(env.base, [l.contiguous() for l in env.levels], env.min_roughness, env.max_roughness))
sa, oa = p.scales.exp(), torch.sigmoid(p.opacities).squeeze(-1).contiguous()
for k, c in enumerate(cams):
    camt = (c.view_matrix.to(dev).contiguous(), c.intrinsic_matrix.to(dev).contiguous(), c.c2w[:, 3].to(dev).contiguous())
    for tight in (False, True):
        fr = F.front_stage(p.means, p.quats, sa, oa, p.normals, p.kd, p.ks, *camt, e, 800, 800, 0.1, 1.0, _MODE["pbr"], tight_tiles=tight)
        torch.cuda.synchronize()
        V, I = int(fr.host_counts[0]), int(fr.host_counts[1])
        r = fr.rects[:V].to(torch.int64)
        x0, y0, x1, y1 = r[:, 0] & 0xffff, r[:, 0] >> 16, r[:, 1] & 0xffff, r[:, 1] >> 16
        nt = (x1 - x0) * (y1 - y0)
        hx = fr.vis[:V, 6]
        print(f"view {k} tight={tight}: V {V} I {I} zero-tile {(nt == 0).sum().item()} ({(nt == 0).float().mean().item():.3f})  hx<0 {(hx < 0).sum().item()}  1 tile {(nt == 1).float().mean().item():.3f}  max {nt.max().item()}")
