#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i
mkdir -p $O
V=geosplatting_amd/build/variants
for tag in base noatomic_raster noglobal_texel nolds_texel notexel; do
  LIBV=""; [ $tag != base ] && LIBV=$V/lib_$tag.so
  GEOSPLAT_LIB=$LIBV timeout 600 python scripts/bench_variant.py --no-cpu-baseline --kernel-iters 6 > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/bench_$tag.json").read())
    print("$tag", round(j["value"],1), "views/s", round(j["ms_per_step"],3), "ms/step; view", round(j["gpu_view_ms_detail"]["graph_ms"],3), "alone", {k: round(v,3) for k,v in j["roofline"]["kernel_ms"].items()}, "in-engine", {k: round(v,3) for k,v in j["roofline"]["kernel_ms_in_engine"].items()})
except Exception as e:
    print("$tag failed", e); print(open("$O/bench_$tag.err").read()[-800:])
PY
done
