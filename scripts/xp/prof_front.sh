#!/bin/bash
# kernel durations of scripts/xp/front_alone.py under rocprofv3
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=/tmp/prof_front
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o r -- python $ROOT/scripts/xp/front_alone.py > $OUT/log.txt 2>&1
cd $ROOT
DB=$(find $OUT -name "*results.db" | head -1)
python scripts/rocprof_summary.py $DB gpurun_out/xp_front.txt
grep -E "front_fwd" gpurun_out/xp_front.txt | cut -c1-130
tail -12 $OUT/log.txt
