#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# A/B of the 7-instruction exact reciprocal in the restart kernel's triangle test (PTAMD_SHORT_RCP=0/1), after the parity suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 90 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED (rc=$?)"; tail -3 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -p timeout --timeout 150 --timeout-method thread > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2 3; do for v in 0 1; do
  echo -n "short_rcp=$v: "; PTAMD_SHORT_RCP=$v timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('value_unpipelined'))" || exit 1
done; done
for v in 0 1; do echo -n "atrium short_rcp=$v: "; PTAMD_SHORT_RCP=$v timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --atrium 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('value_unpipelined'))" || exit 1; done
