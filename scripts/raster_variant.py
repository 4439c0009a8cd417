"""Timing experiment: compositor forward / backward with compile-time variants (diagnostic builds in /tmp)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import geosplatting_amd.build as B
tag = sys.argv[1]
so = f"/tmp/libgeosplat_rv_{tag}.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, *sys.argv[2:], "-shared", "-o", so, *[os.path.join(B.CSRC, s) for s in B.SOURCES]])
import geosplatting_amd._lib as L
L.LIB_PATH = so
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
import io, contextlib
buf = io.StringIO()
import runpy
with contextlib.redirect_stdout(buf):
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(tag, round(d["value"], 1), "views/s", round(d["ms_per_step"], 2), "ms", d["roofline"]["kernel_ms"])
