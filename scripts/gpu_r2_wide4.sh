#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --kernel restart $BARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], r['kernel_ms_per_launch'])"
}
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "trace_rays or large_scene or wide_walk or config4" > $OUT/pytest_wide.log 2>&1; rc=$?; echo "pytest(wide subset) rc=$rc"; tail -3 $OUT/pytest_wide.log
for t in 0 85 341 512; do
  BARGS="--atrium" run "atrium treelet=$t" PTAMD_TREELET=$t
  BARGS="--tessellate 24" run "tessellated treelet=$t" PTAMD_TREELET=$t
done
BARGS="--atrium" run "atrium treelet=341 stack_lds=6" PTAMD_TREELET=341 PTAMD_STACK_LDS=6
BARGS="--atrium --frames-in-flight 1" run "atrium treelet=341 fif=1" PTAMD_TREELET=341
BARGS="--atrium --frames-in-flight 3" run "atrium treelet=341 fif=3" PTAMD_TREELET=341
