#!/bin/bash
# End-of-round ablation: each default of the engine switched off alone (bench.py --no-cpu-baseline, one sample each; base twice)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %7.1f views/s  %7.3f ms/step' % ('$*', d['value'], d['ms_per_step']))"; }
run defaults=1
run GEOSPLAT_TAIL_BATCH=8
run GEOSPLAT_TAIL_BATCH=0
run GEOSPLAT_TAIL_PROJ_STREAM=0
run GEOSPLAT_EARLY_BIN=0
run GEOSPLAT_TIGHT_TILES=0
run GEOSPLAT_BWD_LOG_ORDER=0
run GEOSPLAT_RASTER_LOG=0
run GEOSPLAT_KEY_BITS=32
run GEOSPLAT_FRONT_STREAMS=1
run GEOSPLAT_FRONT=split
run defaults=1
