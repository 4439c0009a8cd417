#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d
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
    print("$tag", round(j["value"],1), "views/s", round(j["ms_per_step"],3), "ms/step; view", j["gpu_view_ms_detail"]["graph_ms"], "in-engine", j["roofline"]["kernel_ms_in_engine"])
except Exception as e:
    print("$tag failed", e); print(open("$O/bench_$tag.err").read()[-1500:])
PY
}
run_bench batch8 GEOSPLAT_TAIL_BATCH=8
run_bench batch4 GEOSPLAT_TAIL_BATCH=4
run_bench batch2 GEOSPLAT_TAIL_BATCH=2
run_bench batch0 GEOSPLAT_TAIL_BATCH=0
run_bench batch8_nopriv GEOSPLAT_TAIL_BATCH=8 GEOSPLAT_TAIL_PRIV=0
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -5 $O/gputest.log
GEOSPLAT_TAIL_BATCH=8 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --kernel-iters 1 > $O/bench_under_rocprof.log 2>&1
DB=$(ls $O/prof/*/*_results.db $O/prof/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $O/r04_d_kernel_stats.txt
rm -rf $O/prof
grep -v "tile_build\|specular_bounds\|specular_kernel\|tile_symmetry\|bounds_tile\|vnormal" $O/r04_d_kernel_stats.txt | head -32 | cut -c1-160
