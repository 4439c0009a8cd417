#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_front.py -x -q -m gpu > $O/gputest_front.log 2>&1; echo "pytest rc=$?" >> $O/gputest_front.log
tail -8 $O/gputest_front.log
run_bench() { # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
j=json.loads(open("$O/bench_$tag.json").read())
print("$tag", round(j["value"],1), "views/s", round(j["ms_per_step"],3), "ms/step; view", j["gpu_view_ms_detail"]["graph_ms"], "in-engine", j["roofline"]["kernel_ms_in_engine"])
PY
}
run_bench fused GEOSPLAT_FRONT=fused
run_bench fused_nopriv GEOSPLAT_FRONT=fused GEOSPLAT_TAIL_PRIV=0
run_bench fused_priv256 GEOSPLAT_FRONT=fused GEOSPLAT_TAIL_PRIV_MAXRES=256
run_bench split GEOSPLAT_FRONT=split
for f in fused; do
  GEOSPLAT_FRONT=$f timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$f -o view -- python scripts/view_kernels_engine.py 7 8 > $O/view_$f.log 2>&1
  DB=$(ls $O/prof_$f/*/*_results.db $O/prof_$f/*_results.db 2>/dev/null | head -1)
  python scripts/rocprof_summary.py $DB $O/r04_view_kernels_alone_$f.txt
  rm -rf $O/prof_$f
  grep -v "tile_build\|specular\|at::native\|tile_symmetry\|bounds_tile\|rocclr\|vnormal\|tile_apply" $O/r04_view_kernels_alone_$f.txt | head -30 | cut -c1-150
done
