#!/bin/bash
# give the engine's streams SLICES of every XCD (hipExtStreamCreateWithCUMask) instead of sharing all CUs.
# usage (GPU box): bash scripts/cu_partition_sweep.sh > gpurun_out/cu_partition_sweep.txt
cd $GRAFT_REPO_ROOT
run() { printf "%-64s" "$1"; env $1 python bench.py --steps 30 --no-cpu-baseline --no-call-shaped --kernel-iters 4 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['roofline'].get('kernel_ms_in_engine') or {}
print(f\"{j['value']:7.1f} views/s  {j['ms_per_step']:6.2f} ms  in-engine fwd {k.get('raster_fwd_kernel',0):.3f} bwd {k.get('raster_bwd_kernel',0):.3f}\")"; }
echo "# CUs lo:hi of EVERY XCD (32 per XCD); unset = all CUs.  bench.py --steps 30, 2 M Gaussians, 800^2, 8 views + prefilter"
run "X=0"
run "GEOSPLAT_CU_SLICES=front=24:32"
run "GEOSPLAT_CU_SLICES=front=20:32"
run "GEOSPLAT_CU_SLICES=front=16:32"
run "GEOSPLAT_CU_SLICES=front=24:32,tail=24:32"
run "GEOSPLAT_CU_SLICES=front=24:32,tail=16:32"
run "GEOSPLAT_CU_SLICES=front=20:32,tail=20:32"
run "GEOSPLAT_CU_SLICES=front=16:32,tail=16:32"
run "GEOSPLAT_CU_SLICES=front=24:32,tail=16:24"
run "GEOSPLAT_CU_SLICES=main=0:24,front=24:32,tail=24:32"
run "GEOSPLAT_CU_SLICES=main=0:24,front=24:32"
run "GEOSPLAT_CU_SLICES=main=0:20,front=20:32,tail=20:32"
run "GEOSPLAT_CU_SLICES=main=0:28,front=28:32,tail=24:32"
run "GEOSPLAT_CU_SLICES=tail=16:32"
run "GEOSPLAT_CU_SLICES=tail=16:32 GEOSPLAT_TAIL_EARLY_BLOCKS=256"
run "X=0"
