#!/bin/bash
# The GPU parity suite under alternative scheduling knobs (ticket size, lane-refill threshold, default kernel).
# PTAMD_DEFAULT_KERNEL=2 is expected to fail the batched-frames test only (the tile kernel cannot batch).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
for v in "PTAMD_TILES_PER_TICKET=2" "PTAMD_REFILL_MIN=16" "PTAMD_REFILL_MIN=1 PTAMD_TILES_PER_TICKET=3" "PTAMD_DEFAULT_KERNEL=5" "PTAMD_DEFAULT_KERNEL=2"; do
  env $v timeout -k 10 300 python -m pytest tests -m gpu -x -q -p timeout --timeout 120 --timeout-method thread > $OUT/pytest_knob.log 2>&1; echo "$v rc=$? $(tail -1 $OUT/pytest_knob.log)"
done
