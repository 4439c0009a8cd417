"""Diagnostic: the compositor's lane statistics on the ENGINE's path (fused front, cull log): compile the library with
-DGS_RASTER_STATS into a scratch .so, run one view of the bench workload through engine.RenderStep, read both counter banks.
python scripts/raster_stats_engine.py [level=7] [out.json]"""
import ctypes as C, hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geosplatting_amd.build as B
so = "/tmp/libgeosplat_stats.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, "-DGS_RASTER_STATS", *os.environ.get("GS_EXTRA_FLAGS", "").split(), "-shared", "-o", so, *[os.path.join(B.CSRC, s) for s in B.SOURCES]])
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
b1, b2, b3 = (C.c_ulonglong * 8)(), (C.c_ulonglong * 8)(), (C.c_ulonglong * 8)()
lib.gs_raster_stats_read(b1, 1); lib.gs_raster_stats2_read(b2, 1); lib.gs_raster_stats3_read(b3, 1)
step([cam], lambda i, img: up, all_reduce=False); torch.cuda.synchronize()
lib.gs_raster_stats_read(b1, 0); lib.gs_raster_stats2_read(b2, 0); lib.gs_raster_stats3_read(b3, 0)
v, w, h = list(b1), list(b2), list(b3)
slots = v[6] * 128
out = {"source_sha16": hashlib.sha256(open(os.path.join(B.CSRC, "gs_raster.hip"), "rb").read()).hexdigest()[:16],
       "workload": f"icosphere level {level}, 800x800, view 0, engine path (fused front, cull log)", "candidates_per_trip": 2,
       "fwd": {"raw_wave_batches": v[0], "culled_records": v[1], "trips": v[3], "valid_pairs": v[2], "dense_batches": w[5], "listed_candidates": w[4]},
       "bwd": {"launched_as": "raster_bwd_log_kernel<3>", "logged_records_consumed": v[5], "trips": v[6], "valid_pairs": v[7],
               "sub_batches": w[1], "popped_candidates": w[0], "rejected_pixel_terminated_earlier": w[2], "rejected_alpha_or_sigma": w[3],
               "candidate_slots": slots, "empty_slots": slots - w[0], "reduction_trips": w[7],
               "walk_trips_by_lanes_with_a_candidate_1-8_9-16_..._57-64": h}}
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
print(f"bwd: {v[6]} trips, {100.0 * v[7] / max(slots, 1):.1f} % of the candidate slots valid, popped {100.0 * w[0] / max(slots, 1):.1f} %, "
      f"{v[5] / max(w[1], 1):.1f} records and {v[6] / max(w[1], 1):.1f} walk trips / {w[7] / max(w[1], 1):.1f} reduction trips per sub-batch")
