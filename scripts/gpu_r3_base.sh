#!/bin/bash
# round 3, first contact: the new full-size textured parity test + the bench line of the unchanged round-2 kernels on this round's box
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "textured_scenes_at_full_size or crate_land" > $OUT/r3_pytest_textured.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/r3_pytest_textured.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > $OUT/r3_bench_base.json 2> $OUT/r3_bench_base.err; echo "bench rc=$?"; cat $OUT/r3_bench_base.json
