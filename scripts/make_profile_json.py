"""profiles/<name>_pmc_traffic.json from the raw per-kernel summary of scripts/pmc_summary.py: the compositor's forward and
backward launches under stable keys, stamped with the sha256[:16] of the kernel source they were measured on (bench.py quotes
the file only while that source is unchanged).  Usage: make_profile_json.py raw.json out.json [commit]"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(sys.argv[1]))
src = os.path.join(ROOT, "geosplatting_amd", "csrc", "gs_raster.hip")
pick = {}
for k, v in raw.items():
    if k.startswith("raster_fwd"):
        pick["raster_fwd_kernel"] = dict(v, launched_as=k)
    if k.startswith("raster_bwd"):
        pick["raster_bwd_kernel"] = dict(v, launched_as=k)
out = {"source_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16], "commit": sys.argv[3] if len(sys.argv) > 3 else "?",
       "workload": "scripts/view_kernels_engine.py 7 (one view through engine.RenderStep, 2M / 800^2; rounds 1-3: scripts/pmc_view.py), rocprofv3 --pmc passes of scripts/run_pmc_r04.sh, mean per launch; "
                   "hbm_bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024",
       "kernels": pick, "all": raw}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print({k: {q: round(w) if isinstance(w, float) else w for q, w in v.items()} for k, v in pick.items()})
