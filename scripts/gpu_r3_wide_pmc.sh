#!/bin/bash
# PMC passes of the wide walk on the atrium for the library in place (or build/libptamd_$1.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
[ -n "$1" ] && cp build/libptamd_$1.so cuda-pathtracer_amd/libptamd.so
PMC_EXTRA=l1x bash scripts/collect_pmc.sh restart --atrium > $OUT/pmc_atrium.log 2>&1; echo "pmc rc=$?"
cp $OUT/pmc_summary_restart.json $OUT/r3_pmc_atrium_${1:-cur}.json
python3 - <<PY
import json
d=json.load(open("$OUT/r3_pmc_atrium_${1:-cur}.json"))
keys=['SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','TCP_TOTAL_CACHE_ACCESSES_sum','TCP_TCC_READ_REQ_sum','TCP_TCC_READ_REQ_LATENCY_sum','TCC_HIT_sum','TCC_MISS_sum','FETCH_SIZE','SQ_WAIT_ANY','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','GRBM_GUI_ACTIVE','SQ_ACTIVE_INST_VALU','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE']
print({k: (round(d[k]/1e6,2) if k in d else None) for k in keys})
print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d['_derived'].items()})
PY
