#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# the gamma step of the resolve pass as a table (default) against pt_powf per channel (PTAMD_GAMMA_TABLE=0): parity both ways, A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 90 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED (rc=$?)"; tail -3 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
for v in 1 0; do
  PTAMD_GAMMA_TABLE=$v timeout -k 10 600 python -m pytest tests -m gpu -x -q -p timeout --timeout 150 --timeout-method thread > $OUT/pytest_gamma_$v.log 2>&1; rc=$?; echo "gamma_table=$v pytest rc=$rc $(tail -1 $OUT/pytest_gamma_$v.log)"
  [ $rc -eq 0 ] || { tail -30 $OUT/pytest_gamma_$v.log; exit 1; }
done
for rep in 1 2 3; do for v in 0 1; do
  echo -n "gamma_table=$v: "; PTAMD_GAMMA_TABLE=$v timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms_per_launch'])" || exit 1
done; done
for v in 0 1; do echo -n "sequential (one at a time) gamma_table=$v: "; PTAMD_GAMMA_TABLE=$v timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --kernel persistent --sequential --frames-in-flight 1 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])" || exit 1; done
