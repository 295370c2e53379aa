#!/bin/bash
# PMC passes (separate runs, --pmc only: no trace domains mixed in) over a short bench run.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
KERNEL=${1:-bvh}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS" \
           "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel $KERNEL > $OUT/pass$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
done
python3 $R/scripts/summarize_pmc.py $OUT $KERNEL > $R/gpurun_out/pmc_summary_$KERNEL.json
FPL=1; if [ "$KERNEL" = "persistent" ]; then FPL=4; fi   # bench.py batches the 4 spp of a frame into one persistent launch
python3 $R/scripts/collect_traffic.py $R/gpurun_out/pmc_summary_$KERNEL.json $KERNEL $R/gpurun_out/traffic_$KERNEL.json $FPL
