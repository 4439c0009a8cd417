"""Diagnostic: compile gs_raster.hip with -DGS_RASTER_STATS into a scratch library and count wave-batches,
ballot survivors and valid lane-pairs of the compositor at the bench workload."""
import ctypes as C, os, subprocess, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import geosplatting_amd.build as B
so = "/tmp/libgeosplat_stats.so"
srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, "-DGS_RASTER_STATS", "-shared", "-o", so, *srcs])
import geosplatting_amd._lib as L
L.LIB_PATH = so
import geosplatting_amd as gs, geosplatting_amd.synthetic as syn
lib = L.lib()
level = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda:0")
sc = syn.sphere_scene(level, seed=1, cubemap_res=64)
cam = syn.blender_cameras(8)[0]
sp = sc.splats.to(dev)
colors = torch.rand(sp.num, 3, device=dev, requires_grad=True)
buf = (C.c_ulonglong * 8)()
lib.gs_raster_stats_read(buf, 1)
lib.gs_raster_stats2_read((C.c_ulonglong * 8)(), 1)
r, a, meta = gs.rasterization(sp.means, sp.quats, sp.scales.exp(), torch.sigmoid(sp.opacities).squeeze(-1), colors,
                              cam.view_matrix.to(dev)[None], cam.intrinsic_matrix.to(dev)[None], 800, 800)
(r.sum() + a.sum()).backward()
torch.cuda.synchronize()
lib.gs_raster_stats_read(buf, 0)
v = list(buf)
buf2 = (C.c_ulonglong * 8)()
lib.gs_raster_stats2_read(buf2, 0)
w = list(buf2)
I = meta["flatten_ids"].numel()
if os.environ.get("GEOSPLAT_RASTER_LANES", "1") != "0":
    if len(sys.argv) > 2:                                  # python scripts/raster_stats.py 7 profiles/r02_raster_stats.json
        import hashlib, json
        src = os.path.join(B.CSRC, "gs_raster.hip")
        json.dump({"source_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16], "workload": f"icosphere level {level}, 800x800, view 0",
                   "I": I, "candidates_per_trip": 2, "fwd": {"raw_wave_batches": v[0], "culled_records": v[1], "trips": v[3], "valid_pairs": v[2]},
                   "bwd": {"raw_wave_batches": v[4], "culled_records": v[5], "trips": v[6], "valid_pairs": v[7],
                           # round 4: what the unused candidate slots of the walk are (second counter bank)
                           "dense_batches": w[1], "popped_candidates": w[0], "rejected_pixel_terminated_earlier": w[2],
                           "rejected_alpha_or_sigma": w[3], "candidate_slots": v[6] * 64 * 2,
                           "empty_slots": v[6] * 64 * 2 - w[0], "sum_of_longest_list_per_batch": w[6], "reduction_trips": w[7]},
                   "fwd_window": {"dense_batches": w[5], "popped_or_listed_candidates": w[4]}},
                  open(sys.argv[2], "w"), indent=1)
    print(f"I={I}  per-lane lists  fwd: raw wave-batches {v[0]}  culled records {v[1]} ({v[1]/max(v[0],1):.1f}/raw batch)  trips {v[3]} "
          f"({64*v[3]/max(v[1],1):.2f} per 64 culled records)  valid pairs {v[2]} ({v[2]/max(v[3],1):.1f}/trip)")
    print(f"       bwd: raw wave-batches {v[4]}  culled records {v[5]} ({v[5]/max(v[4],1):.1f}/raw batch)  trips {v[6]} "
          f"({64*v[6]/max(v[5],1):.2f} per 64 culled records)  valid pairs {v[7]} ({v[7]/max(v[6],1):.1f}/trip)")
    slots = v[6] * 128
    print(f"       bwd slots: {slots} candidate slots in {v[6]} trips; popped {w[0]} ({100 * w[0] / max(slots, 1):.1f} %), of those valid {v[7]} "
          f"({100 * v[7] / max(slots, 1):.1f} % of the slots), rejected: pixel terminated earlier {w[2]}, alpha / sigma {w[3]}; EMPTY {slots - w[0]} "
          f"({100 * (slots - w[0]) / max(slots, 1):.1f} %);  dense batches {w[1]}, longest list per batch {w[6] / max(w[1], 1):.1f}, "
          f"mean list {w[0] / max(w[1], 1) / 64:.1f}, reduction trips per batch {w[7] / max(w[1], 1):.1f}")
    print(f"       fwd window: dense batches {w[5]}, listed candidates {w[4]}")
    sys.exit(0)
print(f"I={I}  fwd: wave-batches {v[0]}  survivors {v[1]} ({v[1]/max(v[0],1):.1f}/batch)  ok lane-pairs {v[2]} ({v[2]/max(v[1],1):.1f}/survivor)")
print(f"       bwd: wave-batches {v[4]}  survivors {v[5]} ({v[5]/max(v[4],1):.1f}/batch)  reduced hits {v[6]}  valid lane-pairs {v[7]} ({v[7]/max(v[6],1):.1f}/hit)")
