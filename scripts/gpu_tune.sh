#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$LABEL', d['config']['kernel'], d['value'], 'Msamples/s', d['ms_per_frame'], 'ms/frame', d['rgba_checksum_rank0_band'])"; }
for cfg in "640 5 1" "768 6 1" "512 6 0" "384 6 1" "1024 8 1"; do set -- $cfg
  rm -f cuda-pathtracer_amd/libptamd.so; make -s lib EXTRA_HIPFLAGS="-DPT_PERSISTENT_THREADS=$1 -DPT_PERSISTENT_WAVES_PER_EU=$2 -DPT_ASM_WALK=$3" 2>&1 | grep -E "error"
  LABEL="pers$1-w$2-asm$3" run --kernel persistent
done
rm -f cuda-pathtracer_amd/libptamd.so; make -s lib 2>&1 | grep error; true
