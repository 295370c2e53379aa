#!/bin/bash
# Kernel time vs band height (the per-GPU share of a multi-GPU split) and frames in flight, headline scene.
# SHARE="--no-share" sizes every launch to the whole GPU (default: 1/n of it with n frames in flight).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 90 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -3 $OUT/smoke.log; exit 1; }
for h in ${HEIGHTS:-8 16 32 64 135 270}; do for f in ${FIF:-1}; do for sh in ${SHARES:-share}; do
opt=""; [ "$sh" = "noshare" ] && opt="--no-share"
timeout -k 10 120 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --height $h --frames-in-flight $f $opt 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('h=$h fif=$f $sh', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms_per_launch'])"
done; done; done
