#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# Round-2 first GPU session: GPU tests (new full-size configs), exec-mask micro-benchmark, the new bench line,
# refill_min A/B on the headline config.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -5 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
timeout -k 10 60 build/exec_half > $OUT/exec_half.log 2>&1; echo "ubench rc=$?"; cat $OUT/exec_half.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
for v in "PTAMD_REFILL_MIN=64" "PTAMD_REFILL_MIN=32" "PTAMD_REFILL_MIN=16" "PTAMD_REFILL_MIN=8"; do
  for fif in 1 2; do
    env $v timeout -k 10 120 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-extra --frames-in-flight $fif 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v fif=$fif', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['box_loop_lane_utilisation'])"
  done
done
