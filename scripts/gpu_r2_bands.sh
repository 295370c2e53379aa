#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['value_unpipelined'], d['value_sequential'], d['ms_per_frame']); print(json.dumps(d['other_configs'],indent=1))"
timeout -k 10 900 python scripts/band_proxy.py --out $OUT/band_proxy.json; echo "proxy rc=$?"
