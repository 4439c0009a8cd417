"""Distribution of the per-tile intersection counts of the bench views (clipped rectangles): sizes the LDS capacity of the tile-local sort."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd import front as F
from geosplatting_amd.engine import params_from_scene
from geosplatting_amd.shading import _MODE, _make_env
dev = torch.device("cuda:0")
level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
scene = syn.sphere_scene(level, seed=1, cubemap_res=64, device=dev)
cams = syn.blender_cameras(num=8, width=800, height=800)
p = params_from_scene(scene, dev)
with torch.no_grad():
    env = gs.as_splitsum(p.cubemap)
e = _make_env(gs.get_fg_lut(dev), gs.TextureSplitSum(env.base, [l.contiguous() for l in env.levels]))
sc, op = p.scales.exp(), torch.sigmoid(p.opacities).squeeze(-1).contiguous()
d = lambda t: t.to(dev, torch.float32).contiguous()
for tight in (True, False):
    for i in (0, 3):
        c = cams[i]
        fr = F.front_stage(p.means, p.quats, sc, op, p.normals, p.kd, p.ks, d(c.view_matrix), d(c.intrinsic_matrix), d(c.c2w[:, 3]), e, 800, 800, 0.1, 1.0,
                           _MODE["pbr"], tight_tiles=tight)
        torch.cuda.synchronize()
        tc = fr.tile_counts.cpu().long()
        nz = tc[tc > 0]
        qs = torch.quantile(nz.float(), torch.tensor([0.5, 0.9, 0.99]))
        print(f"tight={tight} view {i}: I={int(tc.sum())} non-empty tiles {nz.numel()} mean {nz.float().mean():.0f} median {qs[0]:.0f} p90 {qs[1]:.0f} p99 {qs[2]:.0f} max {int(tc.max())}; "
              f"tiles > 2048: {(tc > 2048).sum().item()}, > 4096: {(tc > 4096).sum().item()}, > 8192: {(tc > 8192).sum().item()}")
