for l in "" geosplatting_amd/build/variants/lib_proj256.so geosplatting_amd/build/variants/lib_proj1024.so ""; do
  printf "%-56s" "LIB=$l"; GEOSPLAT_LIB=$l python scripts/bench_variant.py --steps 30 --no-cpu-baseline --kernel-iters 4 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); k=j['roofline'].get('kernel_ms_in_engine') or {}; c=j.get('call_shaped') or {}
print(f\"{j['value']:7.1f} views/s {j['ms_per_step']:6.2f} ms | call-shaped {c.get('views_per_s',0):6.1f} {c.get('ms_per_step',0):6.2f} ms | square {(j.get('value_square_tiles') or {}).get('views_per_s',0):6.1f} | view graph {j['gpu_view_ms_detail'].get('graph_ms')}\")"
done
