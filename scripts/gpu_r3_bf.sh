#!/bin/bash
# branch-free pushes in the four-wide visit (build/libptamd_bf.so) against the previous build (build/libptamd_head.so): wide-walk
# parity on the new library, then the atrium and the tessellated indoor, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
cp build/libptamd_bf.so $LIB
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED"; tail -3 $OUT/smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/r3_bf_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/r3_bf_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $OUT/r3_bf_pytest.log | head; exit 1; }
run() { local v=$1; shift; cp build/libptamd_$v.so $LIB; timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>>$OUT/bf.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', '$*', d['value'])" || exit 1; }
for rep in 1 2 3; do for v in head bf; do run $v --atrium; run $v --tessellate 24; done; done
cp build/libptamd_bf.so $LIB
