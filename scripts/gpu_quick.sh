#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for k in persistent split; do timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --kernel $k 2>>$OUT/bench.err | cut -c1-120; done
