"""profiles/<name>_engine_kernel_ms.json from a rocprofv3 kernel-stats table (scripts/rocprof_summary.py) of the bench command:
the average duration of the compositor kernels INSIDE the three-stream step (where they share the CUs with the front and tail
streams), stamped with the sha of the kernel source.  usage: make_engine_kernel_json.py stats.txt out.json [commit]"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for line in open(sys.argv[1]).read().splitlines()[1:]:
    f = line.split(None, 9)
    if len(f) < 10:
        continue
    rows[f[9]] = {"calls": int(f[0]), "avg_us": float(f[2])}
pick = {}
for name, v in rows.items():
    if "raster_fwd" in name:
        pick["raster_fwd_kernel"] = v["avg_us"] / 1e3
    if "raster_bwd" in name:
        pick["raster_bwd_kernel"] = v["avg_us"] / 1e3
src = os.path.join(ROOT, "geosplatting_amd", "csrc", "gs_raster.hip")
out = {"source_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16], "commit": sys.argv[3] if len(sys.argv) > 3 else "?",
       "workload": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --kernel-iters 1 "
                   "(scripts/run_profile_r03.sh): average kernel duration over the launches of the engine steps, ms",
       "kernel_ms": pick}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out)
