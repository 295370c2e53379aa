#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$LABEL', d['value'], 'Msamples/s', d['ms_per_frame'], 'ms/frame')"; }
for b in 3 5 6; do for rm in 64 16; do LABEL="B$b-refill$rm" PTAMD_REFILL_MIN=$rm run --kernel persistent --bounces $b; done; done
for rm in 64 16; do LABEL="C5-refill$rm" PTAMD_REFILL_MIN=$rm run --kernel persistent --width 3840 --height 2160 --spp 16 --bounces 8 --aperture 0.113 --steps 3 --warmup 1; done
