#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --kernel restart $BARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], r['kernel_ms_per_launch'])"
}
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "trace_rays or large_scene or wide_walk or config4" > $OUT/pytest_wide.log 2>&1; rc=$?; echo "pytest(wide subset) rc=$rc"; tail -3 $OUT/pytest_wide.log
for v in "w6=-DPT_RS4_WAVES_PER_EU=6" "w5=-DPT_RS4_WAVES_PER_EU=5" "w4=-DPT_RS4_WAVES_PER_EU=4"; do
  name=${v%%=*}; flags=${v#*=}
  make -s -B lib EXTRA_HIPFLAGS="$flags" 2>>$OUT/flags.err || { echo "$name: build failed"; continue; }
  BARGS="--atrium" run "$name atrium" X=1
  BARGS="--atrium" run "$name atrium wm4=24" PTAMD_WALK_MIN4=24
  BARGS="--tessellate 24" run "$name tessellated" X=1
done
make -s -B lib 2>>$OUT/flags.err
