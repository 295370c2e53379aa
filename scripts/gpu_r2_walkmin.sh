#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -5 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
PTAMD_WALK_MIN=12 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "restart or fuzz or full_size or config or stats or degenerate or ragged" > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest(walk_min=12) rc=$rc"; tail -4 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
run() { label=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-extra $BARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], r['kernel_ms_per_launch'], r['box_loop_lane_utilisation'])"
}
BARGS="--kernel persistent" run "persistent" X=1
for rm in 12 24; do for wm in 1 4 8 12 16 24 32; do
  BARGS="--kernel restart" run "restart round_min=$rm walk_min=$wm" PTAMD_ROUND_MIN=$rm PTAMD_WALK_MIN=$wm
done; done
PTAMD_WALK_MIN=12 python scripts/gpu_stats.py | tail -6
