#!/bin/bash
# kernels of ONE view alone (scripts/view_kernels_engine.py) under rocprofv3 --kernel-trace --stats -> gpurun_out/<tag>_view_kernels_alone.txt
# usage (on the GPU box, from the repo root): bash scripts/prof_view_alone.sh <tag> [env assignments...]
TAG=${1:-x}; shift
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
env "$@" rocprofv3 --kernel-trace --stats -d $OUT -o r -- python $ROOT/scripts/view_kernels_engine.py 7 8 > $OUT/log.txt 2>&1
cd $ROOT
DB=$(find $OUT -name "*results.db" | head -1)
python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_view_kernels_alone.txt
grep -E "radix|emit|bin|front|build_stream|tile_order|tile_off|raster" gpurun_out/${TAG}_view_kernels_alone.txt | cut -c1-150
tail -2 $OUT/log.txt
rm -rf $OUT
