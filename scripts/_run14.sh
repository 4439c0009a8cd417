cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/run14; mkdir -p $OUT
timeout 1400 python -m pytest tests/test_gpu_front.py tests/test_gpu_shading.py tests/test_gpu_rasterizer.py tests/test_gpu_fullsize.py tests/test_gpu_stage1.py tests/test_gpu_parallel.py -q 2>&1 | tail -6
for v in noglob nolds noatom; do
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/st_$v -o b -- env GEOSPLAT_LIB=geosplatting_amd/build/variants/lib_$v.so python scripts/bench_variant.py --steps 3 --warmup 2 --no-cpu-baseline --kernel-iters 1 > $OUT/b_$v.log 2>&1
DB=$(ls $OUT/st_$v/*/*_results.db $OUT/st_$v/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/stats_$v.txt; echo $v; grep -E "tail_" $OUT/stats_$v.txt | cut -c1-150
rm -rf $OUT/st_$v
done
