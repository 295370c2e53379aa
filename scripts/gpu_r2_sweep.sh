#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# Restart kernel: knob sweep, then compile-time variants (block size / waves per SIMD).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { label=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --kernel restart $BARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], r['kernel_ms_per_launch'], r['box_loop_lane_utilisation'])"
}
for wm in 2 3 4 5 6; do for rm in 8 16 32; do
  run "walk_min=$wm round_min=$rm div=4" PTAMD_WALK_MIN=$wm PTAMD_ROUND_MIN=$rm
done; done
for dv in 2 8; do run "walk_min=4 round_min=16 div=$dv" PTAMD_WALK_MIN=4 PTAMD_ROUND_MIN=16 PTAMD_ROUND_DIV=$dv; done
BARGS="--frames-in-flight 1" run "walk_min=4 round_min=16 fif=1" PTAMD_WALK_MIN=4 PTAMD_ROUND_MIN=16
BARGS="--frames-in-flight 3" run "walk_min=4 round_min=16 fif=3" PTAMD_WALK_MIN=4 PTAMD_ROUND_MIN=16
for v in "t640w5=-DPT_RS_THREADS=640 -DPT_RS_WAVES_PER_EU=5" "t1024w4=-DPT_RS_THREADS=1024 -DPT_RS_WAVES_PER_EU=4" "t768w6=-DPT_RS_THREADS=768 -DPT_RS_WAVES_PER_EU=6" "t256w6=-DPT_RS_THREADS=256 -DPT_RS_WAVES_PER_EU=6"; do
  name=${v%%=*}; flags=${v#*=}
  make -s -B lib EXTRA_HIPFLAGS="$flags" 2>>$OUT/flags.err || { echo "$name: build failed"; continue; }
  run "$name walk_min=4 round_min=16" PTAMD_WALK_MIN=4 PTAMD_ROUND_MIN=16
  BARGS="--frames-in-flight 1" run "$name walk_min=4 round_min=16 fif=1" PTAMD_WALK_MIN=4 PTAMD_ROUND_MIN=16
done
make -s -B lib 2>>$OUT/flags.err
