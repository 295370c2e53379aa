#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -5 $OUT/smoke.log; exit 1; }
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "trace_rays or large_scene or wide_walk" > $OUT/pytest_wide.log 2>&1; rc=$?; echo "pytest(wide subset) rc=$rc"; tail -8 $OUT/pytest_wide.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
run() { label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --atrium --steps 8 --warmup 2 --no-cpu-baseline --no-extra $BARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], r['kernel_ms_per_launch'], r['nodes_per_ray'], r['tris_per_ray'])"
}
BARGS="--kernel persistent" run "atrium persistent(binary walk)" X=1
BARGS="--kernel restart" run "atrium restart(wide) walk_min4=1" X=1
for wm in 4 8 16 24; do BARGS="--kernel restart" run "atrium restart(wide) walk_min4=$wm" PTAMD_WALK_MIN4=$wm; done
BARGS="--kernel restart --frames-in-flight 1" run "atrium restart(wide) fif=1" X=1
BARGS="--kernel restart" run "atrium restart(wide) stack_lds=8" PTAMD_STACK_LDS=8
