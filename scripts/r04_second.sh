#!/bin/bash
# round 4, second GPU call: the fused front / tail -- new parity tests, the whole suite, A/B bench, per-kernel times alone
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_front.py tests/test_gpu_shading.py -x -q -m gpu > $O/gputest_front.log 2>&1; echo "pytest rc=$?" >> $O/gputest_front.log
tail -15 $O/gputest_front.log
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -5 $O/gputest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err
GEOSPLAT_FRONT=split timeout 600 python bench.py --no-cpu-baseline > $O/bench_split.json 2> $O/bench_split.err
for f in fused split; do python - <<PY
import json
j=json.loads(open("$O/bench_$f.json").read())
print("$f", round(j["value"],1), "views/s", round(j["ms_per_step"],3), "ms/step; view alone", j["gpu_view_ms_detail"], "in-engine", j["roofline"]["kernel_ms_in_engine"])
PY
done
for f in fused split; do
  GEOSPLAT_FRONT=$f timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$f -o view -- python scripts/view_kernels_engine.py 7 8 > $O/view_$f.log 2>&1
  DB=$(ls $O/prof_$f/*/*_results.db $O/prof_$f/*_results.db 2>/dev/null | head -1)
  python scripts/rocprof_summary.py $DB $O/r04_view_kernels_alone_$f.txt
  rm -rf $O/prof_$f
  tail -1 $O/view_$f.log
  head -40 $O/r04_view_kernels_alone_$f.txt | cut -c1-150
done
