#!/bin/bash
# round 3: single-GPU proxy of the 8 / 4 / 2-rank split with the round's final build (interleaved 8-row bands, 4 frames in flight)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 900 python scripts/band_proxy.py --ranks 8 4 2 --in-flight 4 --interleave 8 --out $OUT/r03_band_proxy_interleaved8.json; echo "proxy rc=$?"
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('one GPU alone', d['value'], d['ms_per_step'])"
