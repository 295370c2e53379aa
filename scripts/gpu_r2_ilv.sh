#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -5 $OUT/smoke.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python scripts/band_proxy.py --ranks 8 4 --in-flight 2 3 --interleave 16 --out $OUT/band_proxy_interleaved16.json; echo "proxy16 rc=$?"
timeout -k 10 400 python scripts/band_proxy.py --ranks 8 --in-flight 3 --interleave 8 --out $OUT/band_proxy_interleaved8.json; echo "proxy8 rc=$?"
timeout -k 10 400 python scripts/band_proxy.py --ranks 8 --in-flight 3 --interleave 32 --out $OUT/band_proxy_interleaved32.json; echo "proxy32 rc=$?"
