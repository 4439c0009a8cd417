"""Diagnostic: cycles per phase of wave 0 of the LONGEST tile in the forward compositor (-DGS_RASTER_STATS -DGS_RASTER_PHASES build)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geosplatting_amd.build as B
so = "/tmp/libgeosplat_phases.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, "-DGS_RASTER_STATS", "-DGS_RASTER_PHASES", *sys.argv[1:], "-shared", "-o", so,
                       *[os.path.join(B.CSRC, s) for s in B.SOURCES]])
import geosplatting_amd._lib as L
L.LIB_PATH = so
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
lib = L.lib()
dev = torch.device("cuda:0")
sc = syn.sphere_scene(7, seed=1, cubemap_res=64)
cam = syn.blender_cameras(8)[0]
sp = sc.splats.to(dev)
colors = torch.rand(sp.num, 3, device=dev)
buf = (C.c_ulonglong * 8)()
args = (sp.means, sp.quats, sp.scales.exp(), torch.sigmoid(sp.opacities).squeeze(-1), colors, cam.view_matrix.to(dev)[None], cam.intrinsic_matrix.to(dev)[None], 800, 800)
with torch.no_grad():
    gs.rasterization(*args); torch.cuda.synchronize()
    lib.gs_raster_stats_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gs.rasterization(*args); e1.record(); torch.cuda.synchronize()
lib.gs_raster_stats_read(buf, 0)
v = list(buf)
print(f"longest tile, wave 0 (cycle counter ticks): fill {v[0]}  mask+transpose {v[1]}  walk {v[2]}   trips {v[3]}  dense batches {v[4]}   "
      f"(whole rasterization call {e0.elapsed_time(e1):.3f} ms)")
