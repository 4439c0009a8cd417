run() { printf "%-70s" "$1 LIB=$2"; env $1 GEOSPLAT_LIB=$2 python scripts/bench_variant.py --steps 30 --no-cpu-baseline --no-call-shaped --kernel-iters 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['roofline'].get('kernel_ms_in_engine') or {}; a=j['roofline']['kernel_ms']
print(f\"{j['value']:7.1f} views/s {j['ms_per_step']:6.2f} ms | alone fwd {a['raster_fwd_kernel']:.3f} bwd {a['raster_bwd_kernel']:.3f} | in-engine fwd {k.get('raster_fwd_kernel',0):.3f} bwd {k.get('raster_bwd_kernel',0):.3f}\")"; }
V=geosplatting_amd/build/variants
run "X=0" ""
run "X=0" $V/lib_slots528.so
run "GEOSPLAT_RASTER_BLOCKS=5" $V/lib_slots528.so
run "GEOSPLAT_RASTER_BLOCKS=5" $V/lib_occ5.so
run "X=0" $V/lib_occ5.so
run "X=0" ""
