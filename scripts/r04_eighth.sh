#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h
mkdir -p $O
run_bench() { # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --kernel-iters 4 > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/bench_$tag.json").read())
    print("$tag", round(j["value"],1), "views/s", round(j["ms_per_step"],3), "ms/step")
except Exception as e:
    print("$tag failed", e); print(open("$O/bench_$tag.err").read()[-1500:])
PY
}
run_bench base
run_bench first6 GEOSPLAT_TAIL_FIRST=6
run_bench first5 GEOSPLAT_TAIL_FIRST=5
run_bench first7 GEOSPLAT_TAIL_FIRST=7
run_bench base2
timeout 900 python scripts/vertex_mode_errors.py > $O/vertex_mode_errors.txt 2>&1
grep "^seed" $O/vertex_mode_errors.txt | cut -c1-330
# timeline of one step for the main stream
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --kernel-iters 1 > $O/bench_under_rocprof.log 2>&1
DB=$(ls $O/prof/*/*_results.db $O/prof/*_results.db 2>/dev/null | head -1)
python scripts/concurrency_analysis.py $DB > $O/r04_concurrency_one_step.txt 2>&1
rm -rf $O/prof
head -50 $O/r04_concurrency_one_step.txt
