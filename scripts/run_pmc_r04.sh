#!/bin/bash
# PMC passes of ONE view on the engine's path (counters only, no tracing domains besides kernel-trace) -> gpurun_out/pmc_<tag>/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r04}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pass$i -o p -- python scripts/view_kernels_engine.py 7 3 > $OUT/pass$i.log 2>&1
done
python scripts/pmc_summary.py $OUT $OUT/pmc_view.txt $OUT/pmc_traffic_raw.json > /dev/null 2>&1
rm -rf $OUT/pass*/
ls $OUT
