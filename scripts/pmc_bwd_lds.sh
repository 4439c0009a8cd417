#!/bin/bash
# LDS counters of the compositor backward alone (one view through the engine): bash scripts/pmc_bwd_lds.sh <tag> [env ...]
TAG=${1:-x}; shift
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
env "$@" rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT -o p -- python $ROOT/scripts/view_kernels_engine.py 7 3 > $OUT/log.txt 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "raster_bwd" in k or "raster_fwd" in k:
            acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()})
PY
rm -rf $OUT
