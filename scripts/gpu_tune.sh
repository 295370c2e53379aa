#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$LABEL', d['config']['kernel'], d['value'], 'Msamples/s', d['ms_per_frame'], 'ms/frame', d['rgba_checksum_rank0_band'])"; }
for lm in 64 48 32 24 16 8; do
  rm -f cuda-pathtracer_amd/libptamd.so; make -s lib EXTRA_HIPFLAGS="-DPT_LEAF_MIN=${lm}u" 2>&1 | grep -E "error"
  LABEL="leafmin$lm" run --kernel persistent
  timeout -k 10 60 python scripts/gpu_ablate.py 2>&1 | grep "tile       indoor B=4"
done
rm -f cuda-pathtracer_amd/libptamd.so; make -s lib 2>&1 | grep error; true
