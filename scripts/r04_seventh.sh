#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_front.py tests/test_gpu_parallel.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/gputest_sel.log 2>&1; echo "pytest rc=$?" >> $O/gputest_sel.log
tail -6 $O/gputest_sel.log
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
run_bench a
run_bench b
