#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --kernel restart $BARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], r['kernel_ms_per_launch'])"
}
for v in "base=" "dup8=-DPT_EXPERIMENT_DUP_LOADS=8" "dup16=-DPT_EXPERIMENT_DUP_LOADS=16"; do
  name=${v%%=*}; flags=${v#*=}
  make -s -B lib EXTRA_HIPFLAGS="$flags" 2>>$OUT/flags.err || { echo "$name: build failed"; continue; }
  BARGS="--atrium" run "$name atrium" X=1
  BARGS="--atrium" run "$name atrium treelet=0" PTAMD_TREELET=0
done
make -s -B lib 2>>$OUT/flags.err
