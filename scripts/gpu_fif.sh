#!/bin/bash
# Kernel time vs band height (the per-GPU share of a multi-GPU split) and frames in flight, headline scene.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for h in ${HEIGHTS:-8 16 32 64 135 270}; do for f in ${FIF:-1}; do
timeout -k 10 120 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --height $h --frames-in-flight $f 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('h=$h fif=$f', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms_per_launch'])"
done; done
