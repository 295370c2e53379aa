#!/bin/bash
# frames in flight at the driver's 20 steps (and at 100): 1 (the library pipelines), 2 (default), 3
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
for rep in 1 2 3; do for f in 1 2 3; do for st in 20 100; do
  timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps $st --warmup 3 --frames-in-flight $f 2>>$OUT/fif.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('in flight $f steps $st', d['value'])" || exit 1
done; done; done
