#!/bin/bash
# A/B of prebuilt libraries (build/libptamd_<name>.so, built in the work tree before the call): parity suite with the last
# one named, then the headline bench alternating between them.   usage: gpu_ab_libs.sh prev new
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
last=${@: -1}
cp build/libptamd_$last.so $LIB || exit 1
timeout -k 10 90 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED (rc=$?)"; tail -3 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -p timeout --timeout 150 --timeout-method thread > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do for v in "$@"; do
  cp build/libptamd_$v.so $LIB
  timeout -k 10 180 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra $BENCH_ARGS 2>>$OUT/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['roofline']['kernel_ms_per_launch'])" || exit 1
done; done
cp build/libptamd_$last.so $LIB
