#!/bin/bash
# parity suite with the short sqrt / reciprocal of the shading code, then A/B against the full forms (rebuilds per variant)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 90 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED (rc=$?)"; tail -3 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -p timeout --timeout 150 --timeout-method thread > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
bash scripts/gpu_r2_wg.sh "full=-DPT_SHORT_MATH=0" "short=" "full2=-DPT_SHORT_MATH=0" "short2="
