"""bench.py main loop with another build of the library: GEOSPLAT_LIB=<path> python scripts/bench_variant.py [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import geosplatting_amd._lib as L
if os.environ.get("GEOSPLAT_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["GEOSPLAT_LIB"])
import runpy
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
