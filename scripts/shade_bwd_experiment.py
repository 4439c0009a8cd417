"""Timing experiment: shade_bwd with / without the global texel-gradient atomics (diagnostic builds in /tmp)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geosplatting_amd.build as B
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
so = f"/tmp/libgeosplat_{variant}.so"
flags = list(B.FLAGS) + (["-DGS_EXPERIMENT_NO_GLOBAL_TEXEL_ATOMICS"] if variant == "noatom" else []) + sys.argv[2:]
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-shared", "-o", so, *[os.path.join(B.CSRC, s) for s in B.SOURCES]])
import geosplatting_amd._lib as L
L.LIB_PATH = so
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
v = torch.rand_like(col)
for rep in range(3):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); col.backward(v, retain_graph=True); e1.record(); torch.cuda.synchronize()
    print(variant, "shade backward (incl. torch glue) ms:", e0.elapsed_time(e1))
