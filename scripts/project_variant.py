"""Timing experiment: projection forward with compile-time variants (diagnostic builds in /tmp)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geosplatting_amd.build as B
tag = sys.argv[1]
so = f"/tmp/libgeosplat_pv_{tag}.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, *sys.argv[2:], "-shared", "-o", so, *[os.path.join(B.CSRC, s) for s in B.SOURCES]])
import geosplatting_amd._lib as L
L.LIB_PATH = so
import geosplatting_amd.synthetic as syn
from geosplatting_amd.rasterization import _project_stage
dev = torch.device("cuda:0")
sc = syn.sphere_scene(7, seed=1, cubemap_res=64)
cam = syn.blender_cameras(8)[0]
sp = sc.splats.to(dev)
colors = torch.rand(sp.num, 3, device=dev)
args = (sp.means, sp.quats, sp.scales.exp(), torch.sigmoid(sp.opacities).squeeze(-1).contiguous(), colors,
        cam.view_matrix.to(dev).contiguous(), cam.intrinsic_matrix.to(dev).contiguous(), 800, 800, 16, 0.3, 0.01, 1e10, 0.0)
for rep in range(5):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); pr = _project_stage(*args); e1.record(); torch.cuda.synchronize()
print(tag, "project stage ms:", round(e0.elapsed_time(e1), 4), "counts", pr.host_counts.tolist())
