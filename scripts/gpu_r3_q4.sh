#!/bin/bash
# the four-wide walk over 64-byte quantised nodes (PTAMD_WIDE4Q=1) against the float nodes: parity, then the atrium and the
# tessellated indoor at 1080p x 4 spp, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED"; tail -3 $OUT/smoke.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "quantised_four or eight_wide or trace_rays or wide or config4 or large_scene or atrium" > $OUT/r3_q4_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/r3_q4_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $OUT/r3_q4_pytest.log | head -20; exit 1; }
for rep in 1 2; do for q in 0 1; do for args in "--atrium" "--tessellate 24"; do
  PTAMD_TUNING=1 PTAMD_WIDE4Q=$q timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra $args $BENCH_ARGS 2>>$OUT/q4.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('WIDE4Q=$q', '$args', d['value'], d['roofline']['kernel_ms_per_launch'])" || exit 1
done; done; done
