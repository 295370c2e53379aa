#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# Restart kernel bring-up: parity first (smoke with a hard time limit: a kernel that never ends must not take the run with it),
# then A/B against the persistent kernel over the round threshold.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -5 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
run() { # label, env..., -- bench args
  label=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-extra $BARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], r['kernel_ms_per_launch'], r['box_loop_lane_utilisation'])"
}
for fif in 1 2; do
  BARGS="--frames-in-flight $fif --kernel persistent" run "persistent fif=$fif" X=1
  for rm in 4 8 12 16 24 32; do
    BARGS="--frames-in-flight $fif --kernel restart" run "restart min=$rm div=4 fif=$fif" PTAMD_ROUND_MIN=$rm PTAMD_ROUND_DIV=4
  done
  BARGS="--frames-in-flight $fif --kernel restart" run "restart min=24 div=2 fif=$fif" PTAMD_ROUND_MIN=24 PTAMD_ROUND_DIV=2
  BARGS="--frames-in-flight $fif --kernel restart" run "restart min=1 (no cap) fif=$fif" PTAMD_ROUND_MIN=1
done
BARGS="--kernel persistent --width 3840 --height 2160 --spp 16 --bounces 8 --aperture 0.113 --steps 4 --warmup 1" run "c5 persistent" X=1
BARGS="--kernel restart --width 3840 --height 2160 --spp 16 --bounces 8 --aperture 0.113 --steps 4 --warmup 1" run "c5 restart" X=1
BARGS="--kernel persistent --tessellate 24 --steps 6 --warmup 1" run "c4 persistent" X=1
BARGS="--kernel restart --tessellate 24 --steps 6 --warmup 1" run "c4 restart" X=1
