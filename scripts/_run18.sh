cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/run18; mkdir -p $OUT
timeout 1400 python -m pytest tests/test_gpu_front.py tests/test_gpu_fullsize.py tests/test_gpu_stage1.py tests/test_gpu_parallel.py -q 2>&1 | tail -4
for k in 1 0 1 0; do
GEOSPLAT_TAIL_PROJ_STREAM=$k python bench.py --no-cpu-baseline --kernel-iters 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $k', round(d['value'],1), d['ms_per_step'])"
done
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/st -o b -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --kernel-iters 1 > $OUT/b.log 2>&1
DB=$(ls $OUT/st/*/*_results.db $OUT/st/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/stats.txt; grep -E "tail|tile_apply" $OUT/stats.txt | cut -c1-150
python scripts/concurrency_analysis.py $DB 2>&1 | head -14
rm -rf $OUT/st
