#!/bin/bash
# All profile artefacts of round 3 in one call -> gpurun_out/final_r03/ (copied to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final_r03
mkdir -p $OUT
bash scripts/run_profile_r03.sh r03c > $OUT/profile.log 2>&1
cp gpurun_out/prof_r03c/r03c_kernel_stats.txt gpurun_out/prof_r03c/r03c_concurrency_one_step.txt $OUT/
python scripts/make_engine_kernel_json.py $OUT/r03c_kernel_stats.txt $OUT/r03_engine_kernel_ms.json ${1:-?} > $OUT/engine_json.log 2>&1
# one view alone (single stream)
mkdir -p $OUT/view && rocprofv3 --kernel-trace --stats -d $OUT/view/stats -o v -- python scripts/view_kernels.py > $OUT/view/log.txt 2>&1
DB=$(ls $OUT/view/stats/*/*_results.db $OUT/view/stats/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/r03_view_kernels_alone.txt; rm -rf $OUT/view/stats
# PMC passes of one view (compositor traffic) and the lane statistics of the compositor
bash scripts/run_pmc.sh r03 > $OUT/pmc.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_r03 $OUT/r03_pmc_view.txt $OUT/r03_pmc_traffic_raw.json > /dev/null 2>&1
python scripts/make_profile_json.py $OUT/r03_pmc_traffic_raw.json $OUT/r03_pmc_traffic.json ${1:-?} > $OUT/traffic_json.log 2>&1
python scripts/raster_stats.py 7 $OUT/r03_raster_stats.json > $OUT/r03_raster_stats.txt 2>&1
# prefilter alone + its counters
python scripts/prefilter_bench.py > $OUT/r03_prefilter_bench.txt 2>&1
bash scripts/run_pmc_tiles.sh r03 > /dev/null 2>&1
cp gpurun_out/pmc_tiles_r03/summary.txt $OUT/r03_pmc_prefilter_tiles.txt
rm -rf gpurun_out/pmc_r03/pass*/ gpurun_out/pmc_tiles_r03/pass*/ 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT
