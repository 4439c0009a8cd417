#!/bin/bash
# PMC passes over the prefilter apply kernel (scripts/apply_experiment.py <tag> <defs...>) -> gpurun_out/pmc_apply_<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/pmc_apply_$TAG
mkdir -p $OUT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_PERF_SEL_TOTAL_READ_sum TCP_PERF_SEL_TOTAL_HIT_LRU_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pass$i -o p -- python scripts/apply_experiment.py $TAG "$@" > $OUT/pass$i.log 2>&1
done
python - <<PY
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pass*/*counter_collection.csv")):
    per = collections.defaultdict(float); names = {}; grid = {}
    for r in csv.DictReader(open(f)):
        if "specular_apply" not in r["Kernel_Name"]: continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); grid[r["Dispatch_Id"]] = r["Grid_Size"]
    for (d, c), v in per.items():
        vals[grid[d]][c].append(v)
for g in sorted(vals, key=lambda x: -int(x)):
    print("grid", g)
    for c in sorted(vals[g]):
        v = vals[g][c]; print(f"    {c:44s} {sum(v)/len(v):.6g}  (n={len(v)})")
PY
