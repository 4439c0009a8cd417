"""Diagnostic: cycles per phase of wave 0 of block 0 (the LONGEST tile: tiles are launched longest first) of the cull-log backward on
the ENGINE's path (-DGS_RASTER_STATS -DGS_RASTER_PHASES build).  python scripts/raster_phases_engine.py [level=7]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geosplatting_amd.build as B
so = "/tmp/libgeosplat_phases.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, "-DGS_RASTER_STATS", "-DGS_RASTER_PHASES", "-shared", "-o", so, *[os.path.join(B.CSRC, s) for s in B.SOURCES]])
import geosplatting_amd._lib as L
L.LIB_PATH = so
import geosplatting_amd.synthetic as syn
from geosplatting_amd.engine import RenderStep, params_from_scene
lib = L.lib()
level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda:0")
scene = syn.sphere_scene(level, seed=1, cubemap_res=64, device=dev)
cam = syn.blender_cameras(8, 800, 800)[0]
step = RenderStep(params_from_scene(scene, dev), prefilter=False)
up = torch.ones(800, 800, 4, device=dev)
step([cam], lambda i, img: up, all_reduce=False); torch.cuda.synchronize()
step.poll_capacity(wait=True)
b1 = (C.c_ulonglong * 8)()


fw = (C.c_ulonglong * 8)()


class Hook:                                                  # read + clear the counters between the forward and the backward of the view
    def __call__(self, i, img):
        torch.cuda.synchronize(); lib.gs_raster_stats_read(fw, 1)
        return up
lib.gs_raster_stats_read(b1, 1)
step([cam], Hook(), all_reduce=False); torch.cuda.synchronize()
lib.gs_raster_stats_read(b1, 0)
v = list(b1)
tot = v[1] + v[2] + v[3] + v[4]
print(f"BACKWARD (cull log), longest tile, wave 0, cycle-counter ticks: batch setup {v[1]} ({100*v[1]/tot:.0f} %)  walk {v[2]} ({100*v[2]/tot:.0f} %)  "
      f"reduction {v[3]} ({100*v[3]/tot:.0f} %)  commit {v[4]} ({100*v[4]/tot:.0f} %);  walk trips {v[5]}  sub-batches {v[6]}  reduction trips {v[7]}")
print(f"   per walk trip {v[2]/max(v[5],1):.0f} ticks, per reduction trip {v[3]/max(v[7],1):.0f}, per sub-batch: setup {v[1]/max(v[6],1):.0f} commit {v[4]/max(v[6],1):.0f}")
f = list(fw)
ft = f[0] + f[1] + f[2]
print(f"FORWARD (window + cull log), longest tile, wave 0: fill / cull {f[0]} ({100*f[0]/ft:.0f} %)  masks + transpose + log {f[1]} ({100*f[1]/ft:.0f} %)  "
      f"walk {f[2]} ({100*f[2]/ft:.0f} %);  walk trips {f[3]}, {f[2]/max(f[3],1):.0f} ticks per trip")

