for fs in 3 2 3 2; do printf "fronts=$fs  "; GEOSPLAT_SPLAT_FRONT_STREAMS=$fs python bench.py --steps 20 --no-cpu-baseline --kernel-iters 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['call_shaped']; print(round(j['value'],1), round(c['views_per_s'],1), round(c['ms_per_step'],2), j['roofline']['issue'] is not None, j['roofline']['traffic'])"; done
for fs in 3 2; do printf "script fronts=$fs  "; GEOSPLAT_SPLAT_FRONT_STREAMS=$fs CALLSHAPE_NO_SYNC_LOOP=1 python scripts/callshape_step.py 30 2>&1 | grep call-shaped | cut -c1-70; done
