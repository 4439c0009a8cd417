"""Do the front chains of consecutive views overlap when they alternate between streams?  front_stage + bin_stage (+ stream build) of 8
views, nothing else on the GPU: 1 / 2 / 3 / 4 streams.   usage: python scripts/front_overlap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
from geosplatting_amd import front as F
from geosplatting_amd.engine import params_from_scene
from geosplatting_amd.shading import _MODE, _make_env
dev = torch.device("cuda:0")
scene = syn.sphere_scene(7, seed=1, cubemap_res=512, device=dev)
cams = syn.blender_cameras(num=8, width=800, height=800)
p = params_from_scene(scene, dev)
with torch.no_grad():
    env = gs.as_splitsum(p.cubemap)
e = _make_env(gs.get_fg_lut(dev), gs.TextureSplitSum(env.base, [l.contiguous() for l in env.levels]))
sc, op = p.scales.exp(), torch.sigmoid(p.opacities).squeeze(-1).contiguous()
d = lambda t: t.to(dev, torch.float32).contiguous()
camt = [(d(c.view_matrix), d(c.intrinsic_matrix), d(c.c2w[:, 3])) for c in cams]
status = torch.zeros(4, dtype=torch.int64, device=dev)
# learn the capacity once
fr = F.front_stage(p.means, p.quats, sc, op, p.normals, p.kd, p.ks, *camt[0], e, 800, 800, 0.1, 1.0, _MODE["pbr"], tight_tiles=True)
_, V, I = F.bin_stage(fr, None, None)
cap = ((int(I * 1.25) + 65535) // 65536) * 65536
lo = 0xffffffff - int(fr.host_counts[2])
KEY_BITS = int(os.environ.get("KEY_BITS", "24")); KEY_BASE = max(0, lo - (1 << 22)) if KEY_BITS == 24 else 0
def run(n_streams, prepare, reps=10):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    def step():
        keep = []
        for i in range(8):
            s = streams[i % n_streams]
            with torch.cuda.stream(s):
                fr = F.front_stage(p.means, p.quats, sc, op, p.normals, p.kd, p.ks, *camt[i], e, 800, 800, 0.1, 1.0, _MODE["pbr"], KEY_BASE, KEY_BITS, status,
                                   want_packed_index=True, tight_tiles=True)
                keep.append(F.bin_stage(fr, cap, status, prepare=prepare))
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        return keep
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / 8 * 1e3
for prepare in (False, True):
    print(f"front + binning{' + stream build' if prepare else ''}, ms per view: " + "  ".join(f"{n} streams {run(n, prepare):.3f}" for n in (1, 2, 3, 4)))
