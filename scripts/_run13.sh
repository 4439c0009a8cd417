cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/run13; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_front.py tests/test_gpu_shading.py tests/test_gpu_rasterizer.py -x -q 2>&1 | tail -8
for k in pairs loop pairs loop; do
GEOSPLAT_TAIL_KERNEL=$k python bench.py --no-cpu-baseline --kernel-iters 4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', round(d['value'],1), d['ms_per_step'])"
done
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/st -o b -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --kernel-iters 1 > $OUT/b.log 2>&1
DB=$(ls $OUT/st/*/*_results.db $OUT/st/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/stats.txt; grep -E "tail|raster_bwd_log|raster_fwd_window" $OUT/stats.txt | cut -c1-150
python scripts/concurrency_analysis.py $DB 2>&1 | head -12
rm -rf $OUT/st
