"""Experiment: gs_front_fwd alone on the bench scene (scripts/xp/prof_front.sh wraps it in rocprofv3 for kernel-only durations).  python scripts/xp/front_alone.py"""
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
lib = L.lib()
camt = [(c.view_matrix.to(dev).contiguous(), c.intrinsic_matrix.to(dev).contiguous(), c.c2w[:, 3].to(dev).contiguous()) for c in cams]
status = torch.zeros(4, dtype=torch.int64, device=dev)
fr = F.front_stage(p.means, p.quats, sa, oa, p.normals, p.kd, p.ks, *camt[0], e, 800, 800, 0.1, 1.0, _MODE["pbr"], tight_tiles=True)
torch.cuda.synchronize()
rng = F.depth_range(fr.host_counts)
kb = max(0, rng[0] - (1 << 22))
ref = None
for var in [0, 0]:
    ts = []
    for rep in range(12):
        c = camt[rep % 8]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fr = F.front_stage(p.means, p.quats, sa, oa, p.normals, p.kd, p.ks, *c, e, 800, 800, 0.1, 1.0, _MODE["pbr"], key_base=kb, key_bits=24, status=status, tight_tiles=True)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
        if rep == 8:
            if var == 0 and ref is None: ref = (fr.vis.clone(), fr.keys.clone(), fr.rects.clone())
            elif var in (0, 1): print("   identical to base:", torch.equal(ref[0], fr.vis), torch.equal(ref[1], fr.keys), torch.equal(ref[2], fr.rects))
    ts.sort()
    print(f"variant {var:2d}: median {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f}  (setup+front+copy)", flush=True)
