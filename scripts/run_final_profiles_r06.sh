#!/bin/bash
# All profile artefacts of round 6 in one call -> gpurun_out/final_r06/ (copied to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final_r06
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gputest.log 2>&1; echo "pytest rc=$?" >> $OUT/gputest.log
tail -3 $OUT/gputest.log
python bench.py > $OUT/r06_bench_builder.json 2> $OUT/bench.err
cut -c1-300 $OUT/r06_bench_builder.json
# kernel statistics inside the engine step + concurrency of one step
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --kernel-iters 1 > $OUT/bench_under_rocprof.log 2>&1
DB=$(ls $OUT/stats/*/*_results.db $OUT/stats/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/r06_b_final_kernel_stats.txt
python scripts/concurrency_analysis.py $DB > $OUT/r06_concurrency_one_step.txt 2>&1
python scripts/step_boundary_timeline.py $DB > $OUT/r06_step_boundary.txt 2>&1
rm -rf $OUT/stats
# the same step through the reference's call shape (RenderableAttrs.splat + autograd): host vs GPU, kernel stats, per-queue timeline
python scripts/callshape_step.py 20 > $OUT/r06_callshape_host_vs_gpu.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/cs -o cs -- env CALLSHAPE_NO_SYNC_LOOP=1 python scripts/callshape_step.py 8 > $OUT/callshape_under_rocprof.log 2>&1
DB=$(ls $OUT/cs/*/*_results.db $OUT/cs/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/r06_callshape_kernel_stats.txt > /dev/null
python scripts/stream_timeline.py $DB 40 3 > $OUT/r06_callshape_timeline.txt 2>&1
rm -rf $OUT/cs
# one view alone through the engine
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/view -o v -- python scripts/view_kernels_engine.py 7 8 > $OUT/view.log 2>&1
DB=$(ls $OUT/view/*/*_results.db $OUT/view/*_results.db 2>/dev/null | head -1)
python scripts/rocprof_summary.py $DB $OUT/r06_view_kernels_alone.txt; rm -rf $OUT/view
# PMC passes of one view (engine path) and the lane statistics of the compositor
bash scripts/run_pmc_r04.sh r06 > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_r06/pmc_view.txt $OUT/r06_pmc_view.txt
python scripts/make_profile_json.py gpurun_out/pmc_r06/pmc_traffic_raw.json $OUT/r06_pmc_traffic.json ${1:-?} > $OUT/traffic_json.log 2>&1
timeout 600 python scripts/raster_stats_engine.py 7 $OUT/r06_raster_stats.json > $OUT/r06_raster_stats.txt 2>&1
timeout 600 python scripts/prefilter_bench.py > $OUT/r06_prefilter_bench.txt 2>&1
timeout 600 python scripts/host_ahead.py 20 > $OUT/r06_host_ahead.txt 2>&1
timeout 600 python scripts/soak.py 1000 > $OUT/r06_soak.txt 2>&1
timeout 600 python scripts/soak_callshape.py 500 > $OUT/r06_soak_callshape.txt 2>&1
# two ranks on the one GPU (gloo): both scaling modes of bench.py end to end -- never a measurement
export GEOSPLAT_DEBUG_SHARE_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --settle-seconds 0 > $OUT/r06_bench_2rank_strong_debug.log 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --weak --steps 3 --warmup 1 --no-cpu-baseline --settle-seconds 0 > $OUT/r06_bench_2rank_weak_debug.log 2>&1
tail -1 $OUT/r06_bench_2rank_strong_debug.log | cut -c1-200; tail -1 $OUT/r06_bench_2rank_weak_debug.log | cut -c1-200
ls -la $OUT
