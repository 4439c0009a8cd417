#!/bin/bash
# rocprofv3 kernel statistics of the stage-1 iteration (scripts/stage1_scale.py: BASELINE config 5's size by default) -> gpurun_out/prof_stage1/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_stage1
R=${1:-208}
IT=${2:-12}
mkdir -p $OUT
python scripts/stage1_scale.py $R $IT > $OUT/plain.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s1 -- python scripts/stage1_scale.py $R $IT > $OUT/under_rocprof.log 2>&1
DB=$(ls $OUT/stats/*/*_results.db $OUT/stats/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/stage1_kernel_stats.txt
rm -rf $OUT/stats
