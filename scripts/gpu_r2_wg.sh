#!/bin/bash
# restart kernel: workgroup shapes / register budgets, rebuilt per variant ("name=flags" arguments; no arguments: the set below)
#   2 workgroups x 12 waves (default) against 1 workgroup x 16 waves per CU (one scene copy, 108 KB of LDS left) and 2 x 8
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
[ $# -gt 0 ] || set -- "default=" "wg1024x4=-DPT_RS_THREADS=1024 -DPT_RS_WAVES_PER_EU=4" "wg512x4=-DPT_RS_THREADS=512 -DPT_RS_WAVES_PER_EU=4" "default2="
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}
  make -s -B lib EXTRA_HIPFLAGS="$flags" 2>>$OUT/wg.err || { echo "$name: build failed"; continue; }
  for rep in 1 2; do
    timeout -k 10 180 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>>$OUT/wg.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name', d['value'], d['roofline']['kernel_ms_per_launch'])" || { echo "$name: bench failed"; break; }
  done
done
make -s -B lib 2>>$OUT/wg.err
