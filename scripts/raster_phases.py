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
NT = 2500
tl = (C.c_ulonglong * (3 * NT))()
lib.gs_raster_timeline_read.argtypes = [C.c_void_p, C.c_int, C.c_int]


def timeline(tag):
    """Per-block start/end (100 MHz wall clock): where does the launch's time go?"""
    import numpy as np
    lib.gs_raster_timeline_read(tl, NT, 1)
    a = np.array(list(tl), dtype=np.uint64).reshape(NT, 3)
    st = (~a[:, 0]).astype(np.int64); en = a[:, 1].astype(np.int64); ln = a[:, 2].astype(np.int64)
    t0 = st.min()
    st = (st - t0) / 100.0; en = (en - t0) / 100.0          # microseconds
    dur = en - st
    print(f"{tag}: launch span {en.max():.1f} us; block durations: max {dur.max():.1f} mean {dur.mean():.1f} sum {dur.sum() / 1000:.2f} ms "
          f"(= {dur.sum() / en.max():.0f} blocks busy on average of {256 * 4} slots)")
    order = np.argsort(-dur)[:8]
    print("   longest blocks (LPT position, list length, start, duration): " + ", ".join(f"#{i} n={ln[i]} @{st[i]:.0f} {dur[i]:.0f}us" for i in order))
    last = np.argsort(-en)[:8]
    print("   last to finish: " + ", ".join(f"#{i} n={ln[i]} @{st[i]:.0f} {dur[i]:.0f}us" for i in last))
    for lo, hi in ((0, 100), (100, 300), (300, 600), (600, 1200), (1200, 2500)):
        d = dur[lo:hi]; n = ln[lo:hi]
        print(f"   LPT {lo:4d}-{hi:4d}: list {n.mean():7.0f}  duration mean {d.mean():6.1f} max {d.max():6.1f} us  ns/record {1000 * d.sum() / max(n.sum(), 1):.1f}  start {st[lo:hi].mean():.0f}")

args = (sp.means, sp.quats, sp.scales.exp(), torch.sigmoid(sp.opacities).squeeze(-1), colors, cam.view_matrix.to(dev)[None], cam.intrinsic_matrix.to(dev)[None], 800, 800)
with torch.no_grad():
    gs.rasterization(*args); torch.cuda.synchronize()
    lib.gs_raster_stats_read(buf, 1); lib.gs_raster_timeline_read(tl, NT, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gs.rasterization(*args); e1.record(); torch.cuda.synchronize()
lib.gs_raster_stats_read(buf, 0)
timeline("forward")
v = list(buf)
print(f"FORWARD  longest tile, wave 0 (cycle counter ticks): fill {v[0]}  mask+transpose {v[1]}  walk {v[2]}   trips {v[3]}  dense batches {v[4]}   "
      f"(whole rasterization call {e0.elapsed_time(e1):.3f} ms)")

# backward: same tile, wave 0 of raster_bwd_lanes2_kernel
m = [t.clone().requires_grad_(True) for t in args[:5]]
out = gs.rasterization(*m, *args[5:])
torch.cuda.synchronize(); lib.gs_raster_stats_read(buf, 1); lib.gs_raster_timeline_read(tl, NT, 1)
out[0].sum().backward(); torch.cuda.synchronize()
lib.gs_raster_stats_read(buf, 0)
timeline("backward")
v = list(buf)
print(f"BACKWARD longest tile, wave 0: fill {v[0]}  mask+scan+transpose {v[1]}  walk {v[2]}  reduce {v[3]}  stage+commit {v[4]}   "
      f"walk trips {v[5]}  reduce trips {v[7]}  dense batches {v[6]}")
