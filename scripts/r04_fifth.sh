#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_front.py -x -q -m gpu > $O/gputest_front.log 2>&1; echo "pytest rc=$?" >> $O/gputest_front.log
tail -12 $O/gputest_front.log
run_bench() { # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/bench_$tag.json").read())
    print("$tag", round(j["value"],1), "views/s", round(j["ms_per_step"],3), "ms/step; view", j["gpu_view_ms_detail"]["graph_ms"], "alone", j["roofline"]["kernel_ms"], "in-engine", j["roofline"]["kernel_ms_in_engine"])
except Exception as e:
    print("$tag failed", e); print(open("$O/bench_$tag.err").read()[-1500:])
PY
}
run_bench log GEOSPLAT_RASTER_LOG=1
run_bench nolog GEOSPLAT_RASTER_LOG=0
run_bench log2 GEOSPLAT_RASTER_LOG=1
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -5 $O/gputest.log
