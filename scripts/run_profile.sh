#!/bin/bash
# rocprofv3 kernel statistics of the bench command + PMC passes of one view -> gpurun_out/prof_<tag>/ (copy the summaries to profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
DB=$(ls $OUT/stats/*/*_results.db $OUT/stats/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/${TAG}_kernel_stats.txt
bash scripts/run_pmc.sh $TAG > $OUT/pmc.log 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_$TAG $OUT/${TAG}_pmc_view.txt $OUT/${TAG}_pmc_traffic_raw.json > /dev/null 2>&1
ls $OUT
