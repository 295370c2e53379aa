#!/bin/bash
# LDS-typed pointers for the wide walk's stack and treelet (build/libptamd_as.so: no FLAT loads / stores left) against the previous
# build (build/libptamd_head.so): wide-walk parity on the new library, then the atrium and the tessellated indoor, alternating;
# float nodes too (PTAMD_WIDE4Q=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
cp build/libptamd_as.so $LIB
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED"; tail -3 $OUT/smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/r3_as_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/r3_as_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $OUT/r3_as_pytest.log | head; exit 1; }
run() { local v=$1 q=$2; shift 2; cp build/libptamd_$v.so $LIB; PTAMD_TUNING=1 PTAMD_WIDE4Q=$q timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>>$OUT/as.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v WIDE4Q=$q', '$*', d['value'])" || exit 1; }
for rep in 1 2; do for v in head as; do
  run $v 1 --atrium; run $v 1 --tessellate 24; run $v 0 --atrium
done; done
cp build/libptamd_as.so $LIB
