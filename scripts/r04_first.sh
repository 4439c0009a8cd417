#!/bin/bash
# round 4, first GPU call: parity suite on the changed engine, the new compositor statistics, the bench line with its new fields,
# and the two-rank (one shared GPU, gloo: never a measurement) logs of both scaling modes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
timeout 600 python scripts/raster_stats.py 7 $O/r04_raster_stats.json > $O/raster_stats.txt 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
export GEOSPLAT_DEBUG_SHARE_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --settle-seconds 0 > $O/bench_2rank_strong_debug.log 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --weak --steps 3 --warmup 1 --no-cpu-baseline --settle-seconds 0 > $O/bench_2rank_weak_debug.log 2>&1
tail -3 $O/gputest.log; cat $O/raster_stats.txt | tail -5; cut -c1-600 $O/bench_default.json; tail -2 $O/bench_2rank_strong_debug.log | cut -c1-400; tail -2 $O/bench_2rank_weak_debug.log | cut -c1-400
