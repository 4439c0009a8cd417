#!/bin/bash
# rocprofv3 kernel trace of the call-shaped step (scripts/callshape_step.py) -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r05_callshape}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python scripts/callshape_step.py 20 > $OUT/host_vs_gpu.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o cs -- env CALLSHAPE_NO_SYNC_LOOP=1 python scripts/callshape_step.py 8 > $OUT/under_rocprof.log 2>&1
DB=$(ls $OUT/stats/*/*_results.db $OUT/stats/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/${TAG}_kernel_stats.txt > /dev/null
python scripts/stream_timeline.py $DB 40 4 > $OUT/${TAG}_timeline.txt 2>&1
cat $OUT/host_vs_gpu.txt; cat $OUT/${TAG}_timeline.txt
