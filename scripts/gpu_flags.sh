#!/bin/bash
# A/B of compiler flag sets on the GPU box: rebuilds libptamd.so per variant and runs the headline bench.
# usage: gpu_flags.sh "name1=flags1" "name2=flags2" ...   (flags may be empty)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
: > $OUT/flags.log
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}
  make -s -B lib EXTRA_HIPFLAGS="$flags" 2>>$OUT/flags.err || { echo "$name: build failed" | tee -a $OUT/flags.log; continue; }
  for rep in 1 2; do
    line=$(timeout -k 10 180 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>>$OUT/flags.err) || { echo "$name: bench failed" | tee -a $OUT/flags.log; break; }
    echo "$name $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" | tee -a $OUT/flags.log
  done
done
# leave the default build in place and check parity with it
make -s -B lib 2>>$OUT/flags.err
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
