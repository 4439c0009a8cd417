for fs in 1 2; do printf "fronts=$fs  "; GEOSPLAT_SPLAT_FRONT_STREAMS=$fs python bench.py --steps 20 --no-cpu-baseline --kernel-iters 2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j['call_shaped']; print(round(j['value'],1), round(c['views_per_s'],1), round(c['ms_per_step'],2))"; done
for fs in 1 2; do printf "script fronts=$fs  "; GEOSPLAT_SPLAT_FRONT_STREAMS=$fs CALLSHAPE_NO_SYNC_LOOP=1 python scripts/callshape_step.py 30 2>&1 | grep call-shaped | cut -c1-70; done
for tb in "8" "4" "2" "1" "3,3,2" ; do printf "script TAIL_BATCH=$tb  "; GEOSPLAT_TAIL_BATCH=$tb CALLSHAPE_NO_SYNC_LOOP=1 python scripts/callshape_step.py 30 2>&1 | grep call-shaped | cut -c1-70; done
