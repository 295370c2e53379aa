#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# escaped paths parked and finished in their own pass: parity, then A/B against the build without it and against itself switched off
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
cp build/libptamd_park.so $LIB
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED"; tail -3 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/r3_pytest_park.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/r3_pytest_park.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do for v in "w4:X=0" "park:PTAMD_PARK_ESCAPED=1" "park:PTAMD_PARK_ESCAPED=0"; do
  lib=${v%%:*}; envs=${v#*:}
  cp build/libptamd_$lib.so $LIB
  env $envs timeout -k 10 180 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['roofline']['kernel_ms_per_launch'])" || exit 1
done; done
for v in "w4:X=0" "park:PTAMD_PARK_ESCAPED=1"; do
  lib=${v%%:*}; envs=${v#*:}
  cp build/libptamd_$lib.so $LIB
  env $envs timeout -k 10 180 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --bounces 8 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('8 bounces $v', d['value'], d['roofline']['kernel_ms_per_launch'])" || exit 1
done
cp build/libptamd_park.so $LIB
