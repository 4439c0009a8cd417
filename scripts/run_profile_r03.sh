#!/bin/bash
# rocprofv3 kernel statistics + three-stream concurrency of the bench command -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r03a}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --kernel-iters 1 > $OUT/bench_under_rocprof.log 2>&1
DB=$(ls $OUT/stats/*/*_results.db $OUT/stats/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/${TAG}_kernel_stats.txt
python scripts/concurrency_analysis.py $DB > $OUT/${TAG}_concurrency_one_step.txt 2>&1
rm -rf $OUT/stats
ls $OUT
