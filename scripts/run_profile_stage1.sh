#!/bin/bash
# rocprofv3 kernel statistics of the stage-1 iteration at BASELINE config 5's size (scripts/stage1_scale.py) -> gpurun_out/prof_stage1/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_stage1
mkdir -p $OUT
python scripts/stage1_scale.py 208 8 > $OUT/plain.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s1 -- python scripts/stage1_scale.py 208 6 > $OUT/under_rocprof.log 2>&1
DB=$(ls $OUT/stats/*/*_results.db $OUT/stats/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/stage1_kernel_stats.txt
rm -rf $OUT/stats
