#!/bin/bash
# Runtime-knob sweep of the persistent kernel on the GPU box (headline config unless BENCH_ARGS says otherwise).
# usage: gpu_knobs.sh "ENV1=a ENV2=b" "ENV1=c" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 90 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -3 $OUT/smoke.log; exit 1; }
for v in "$@"; do
  for rep in 1 2; do
    env $v timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
done
