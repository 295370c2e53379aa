#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
# smoke first and under a short limit: a kernel that never ends must not take the whole test run with it
timeout -k 10 90 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung (rc=$?)"; tail -3 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 400 python -m pytest tests -m gpu -x -q -p timeout --timeout 120 --timeout-method thread > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for k in persistent split; do timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k 2>>$OUT/bench.err | cut -c1-120; done
