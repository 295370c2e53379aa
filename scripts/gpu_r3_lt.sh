#!/bin/bash
# light spheres as leaf records (build/libptamd_lt.so; PTAMD_LIGHT_LEAVES=0 on it: the light loop) against the previous build
# (build/libptamd_head.so): the whole GPU suite on the new library, then alternating runs of the bench configurations
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
cp build/libptamd_lt.so $LIB
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED"; tail -3 $OUT/smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/r3_lt_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/r3_lt_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $OUT/r3_lt_pytest.log | head; exit 1; }
run() { local v=$1 k=$2; shift 2; cp build/libptamd_$v.so $LIB; PTAMD_TUNING=1 PTAMD_LIGHT_LEAVES=$k timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline "$@" 2>>$OUT/lt.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v leaves=$k', '$*', d['value'])" || exit 1; }
for rep in 1 2 3; do
  run head 0 --steps 40; run lt 1 --steps 40; run lt 0 --steps 40
done
for rep in 1 2; do for cfg in "head 0" "lt 1"; do set -- $cfg
  run $1 $2 --steps 6 --width 3840 --height 2160 --spp 16 --bounces 8 --aperture 0.113
  run $1 $2 --steps 20 --fix-backslashes
  run $1 $2 --steps 20 --scene assets/crate_land.scene
done; done
cp build/libptamd_lt.so $LIB
