#!/bin/bash
# PMC passes (separate runs, --pmc only: no trace domains mixed in) over a short headline-only bench run
# (`--no-extra`: every profiled megakernel dispatch is a launch of the headline configuration).
#   bash scripts/collect_pmc.sh [kernel] [extra bench args...]   ->  gpurun_out/pmc_summary_<kernel>.json, gpurun_out/pmc_latest.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
KERNEL=${1:-persistent}
shift
mkdir -p $OUT
rm -rf $OUT/pass*
cd /tmp && export TMPDIR=/tmp
i=0
SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" \
      "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
      "TCC_HIT_sum TCC_MISS_sum")
# PMC_EXTRA=l1: two more passes for walks served from the caches (vector L1 / texture-addresser counters)
if [ "$PMC_EXTRA" = "l1x" ]; then
  SETS+=("TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum")
fi
# PMC_EXTRA=ifetch: instruction supply (fetch requests of the waves, the instruction cache shared by two CUs, instruction mix)
if [ "$PMC_EXTRA" = "ifetch" ]; then
  SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_ANY" \
        "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_STALL" \
        "SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM GRBM_GUI_ACTIVE")
fi
# PMC_EXTRA=tcp: busy clocks of the vector L1 (gate enable) and its tag lookups.  (A second pass with the stall counters
# TCP_TCP_TA_DATA_STALL_CYCLES / TCP_TCR_TCP_STALL_CYCLES / TCP_TD_TCP_STALL_CYCLES / TCP_READ_TAGCONFLICT_STALL_CYCLES hung rocprofv3
# on this pool in round 4, as the TA_* set does: do not add them.)
if [ "$PMC_EXTRA" = "tcp" ]; then
  SETS=("GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum")
fi
if [ "$PMC_EXTRA" = "l1" ]; then
  SETS+=("TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum")
fi
for SET in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --kernel $KERNEL "$@" > $OUT/pass$i.log 2>&1
  echo "pass $i ($SET) rc=$?"
done
# PMC_OUT: name of the small record bench.py reads (default pmc_latest.json = the headline; other workloads: pmc_atrium.json, ...)
python3 $R/scripts/summarize_pmc.py $OUT $KERNEL $R/gpurun_out/${PMC_OUT:-pmc_latest.json} "$@" > $R/gpurun_out/pmc_summary_$KERNEL.json
