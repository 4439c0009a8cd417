#!/bin/bash
# PMC passes over the tiled prefilter kernels (scripts/prefilter_bench.py) -> gpurun_out/pmc_tiles_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-a}
OUT=gpurun_out/pmc_tiles_$TAG
mkdir -p $OUT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pass$i -o p -- python scripts/prefilter_bench.py ${2:-512} > $OUT/pass$i.log 2>&1
done
python - <<PY > $OUT/summary.txt
import csv, glob, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/pass*/*counter_collection.csv")):
    per = collections.defaultdict(float); key = {}
    for r in csv.DictReader(open(f)):
        if "tile_apply" not in r["Kernel_Name"]: continue
        k = ("bwd" if "ILb1E" in r["Kernel_Name"] else "fwd") + " grid " + r["Grid_Size"] + " lds " + r.get("LDS_Block_Size", "?")
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); key[r["Dispatch_Id"]] = k
    for (d, c), v in per.items():
        vals[key[d]][c].append(v)
for g in sorted(vals):
    print(g)
    for c in sorted(vals[g]):
        v = vals[g][c]; print(f"    {c:44s} {sum(v)/len(v):.6g}  (n={len(v)})")
PY
cat $OUT/summary.txt
