"""Timing experiment: compositor backward with / without its per-hit atomics (diagnostic build in /tmp)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geosplatting_amd.build as B
variant = sys.argv[1] if len(sys.argv) > 1 else "base"
so = f"/tmp/libgeosplat_rb_{variant}.so"
flags = list(B.FLAGS) + (["-DGS_EXPERIMENT_NO_RASTER_ATOMICS"] if variant == "noatom" else [])
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-shared", "-o", so, *[os.path.join(B.CSRC, s) for s in B.SOURCES]])
import geosplatting_amd._lib as L
L.LIB_PATH = so
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
dev = torch.device("cuda:0")
sc = syn.sphere_scene(7, seed=1, cubemap_res=64)
cam = syn.blender_cameras(8)[0]
sp = sc.splats.to(dev)
colors = torch.rand(sp.num, 3, device=dev, requires_grad=True)
r, a, meta = gs.rasterization(sp.means, sp.quats, sp.scales.exp(), torch.sigmoid(sp.opacities).squeeze(-1), colors,
                              cam.view_matrix.to(dev)[None], cam.intrinsic_matrix.to(dev)[None], 800, 800)
v = torch.rand_like(r)
for rep in range(4):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); r.backward(v, retain_graph=True); e1.record(); torch.cuda.synchronize()
    print(variant, "rasterization backward ms:", e0.elapsed_time(e1))
