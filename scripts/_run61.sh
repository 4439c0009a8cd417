cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { v=$1; shift; env "$@" python bench.py --no-cpu-baseline --views-total $v 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('views=$v %-28s %7.1f views/s  %7.3f ms/step' % ('$*', d['value'], d['ms_per_step']))"; }
for r in 1 2; do
run 4 A=1
run 4 GEOSPLAT_TAIL_BATCH=2,1,1
run 4 GEOSPLAT_TAIL_BATCH=1
run 3 A=1
run 3 GEOSPLAT_TAIL_BATCH=1
done
