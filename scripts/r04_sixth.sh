#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f
mkdir -p $O
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
run_bench hist1 GEOSPLAT_FRONT_HIST=1
run_bench hist0 GEOSPLAT_FRONT_HIST=0
run_bench front1 GEOSPLAT_FRONT_STREAMS=1
run_bench front3 GEOSPLAT_FRONT_STREAMS=3
timeout 600 python scripts/raster_stats_engine.py 7 $O/r04_raster_stats.json > $O/raster_stats.txt 2>&1
tail -3 $O/raster_stats.txt
bash scripts/run_pmc_r04.sh r04a > $O/pmc.log 2>&1
grep -A12 "raster_bwd_log\|raster_fwd_window\|tail_multi\|front_fwd" gpurun_out/pmc_r04a/pmc_view.txt | head -90
cp gpurun_out/pmc_r04a/pmc_view.txt $O/r04_pmc_view.txt; cp gpurun_out/pmc_r04a/pmc_traffic_raw.json $O/
