"""Compile libgeosplat_hip with extra -D flags into geosplatting_amd/build/variants/lib_<name>.so (travels to the GPU box with
gpurun; git-ignored).  usage: python scripts/build_variant.py <name> [-DFLAG ...];  run with GEOSPLAT_LIB=<path> (scripts/raster_ab.py)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import geosplatting_amd.build as B
name, flags = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(B.HERE, "build", "variants")
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, f"lib_{name}.so")
srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *B.FLAGS, *flags, "-shared", "-o", so, *srcs])
print(so)
