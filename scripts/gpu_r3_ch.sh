#!/bin/bash
# box loop variants against each other: build/libptamd_<v>.so, $NEW = the candidate (whole GPU suite first), $OLD = what it replaces;
# alternating headline runs on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
NEW=${NEW:-ch}; OLD=${OLD:-w4}
LIB=cuda-pathtracer_amd/libptamd.so
cp build/libptamd_$NEW.so $LIB
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED"; tail -3 $OUT/smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/r3_ch_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/r3_ch_pytest.log
[ $rc -eq 0 ] || exit 1
PTAMD_TUNING=1 PTAMD_DEFAULT_KERNEL=2 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "eight_wide" 2>&1 | tail -1
for rep in 1 2 3; do for v in $OLD $NEW; do
  cp build/libptamd_$v.so $LIB
  timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 3 2>>$OUT/ch.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['roofline']['kernel_ms_per_launch'])" || exit 1
done; done
cp build/libptamd_$NEW.so $LIB
