#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# The GPU parity suite under alternative scheduling knobs (ticket size, refill / round / walk thresholds, default kernel,
# stack and treelet sizes of the four-wide walk, pools in global memory, full division in the triangle test, pt_powf instead of the gamma table).  Under a pinned default kernel that cannot batch frames
# (PTAMD_DEFAULT_KERNEL=1/2/4) the batched cases skip themselves (tests/test_gpu_parity.py: batched_ok).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R; i=0
for v in "PTAMD_TILES_PER_TICKET=2" "PTAMD_REFILL_MIN=1 PTAMD_TILES_PER_TICKET=3 PTAMD_DEFAULT_KERNEL=3" \
         "PTAMD_ROUND_MIN=1 PTAMD_WALK_MIN=1" "PTAMD_ROUND_MIN=64 PTAMD_ROUND_DIV=1 PTAMD_WALK_MIN=40 PTAMD_WALK_MIN4=40" \
         "PTAMD_ROUND_MIN=3 PTAMD_WALK_MIN=2 PTAMD_WALK_MIN4=2 PTAMD_TREELET=0 PTAMD_STACK_LDS=3" "PTAMD_TREELET=1000 PTAMD_TILES_PER_TICKET=5" "PTAMD_POOL_LDS=0 PTAMD_SHORT_RCP=0 PTAMD_GAMMA_TABLE=0" \
         "PTAMD_OVERLAP=0" "PTAMD_XCD_REGIONS=2 PTAMD_ROUND_DIV=7" "PTAMD_XCD_REGIONS=1 PTAMD_POOL_LDS_WIDE=1 PTAMD_ROUND_DIV=3 PTAMD_ROUND_MIN=5" "PTAMD_WIDE4Q=1" "PTAMD_DEFAULT_KERNEL=5" "PTAMD_DEFAULT_KERNEL=2"; do
  env $v timeout -k 10 600 python -m pytest tests -m gpu -x -q -p timeout --timeout 200 --timeout-method thread > "$OUT/pytest_knob_$i.log" 2>&1; echo "$v rc=$? $(tail -1 "$OUT/pytest_knob_$i.log")"; i=$((i+1))
done
