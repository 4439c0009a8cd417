"""Build libgeosplat_hip.so (gfx950) in-tree with hipcc.  `python -m geosplatting_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgeosplat_hip.so")
SOURCES = ["gs_project.hip", "gs_sort.hip", "gs_raster.hip", "gs_shade.hip", "gs_front.hip", "gs_splitsum.hip", "gs_splitsum_tiles.hip", "gs_mesh.hip", "gs_loss.hip", "gs_hashgrid.hip", "gs_flexicubes.hip"]
HEADERS = ["gs_common.h", "gs_cube.h", "gs_project_dev.h", "gs_shade_dev.h", "gs_splitsum_math.h", "gs_tone.h", os.path.join("..", "..", "include", "geosplat_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-ffp-contract=fast-honor-pragmas", "-Wno-unused-result",
         # one lane commits per-Gaussian sums: the wave-uniform atomic optimiser (mbcnt/bcnt/mul per atomic) only adds work
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not (force or _stale()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
