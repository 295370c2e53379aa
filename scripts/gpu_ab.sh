#!/bin/bash
# A/B of prebuilt libraries (build/libptamd_<name>.so, scripts/build_variants.sh) with per-variant tuning knobs, alternating on ONE box.
#   VARIANTS="a b:PTAMD_XCD_REGIONS=0,PTAMD_TREELET=256"  BENCH_ARGS="--atrium"  PARITY="a b"  PYTEST_K="wide or trace_rays"  REPS=2  gpu_ab.sh
# PARITY: libraries that run the GPU parity tests first (PYTEST_K selects; empty = the whole suite).  Knobs need PTAMD_TUNING=1 (exported here).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
export PTAMD_TUNING=1
cp $LIB build/libptamd_default.so 2>/dev/null
for v in $PARITY; do
  cp build/libptamd_$v.so $LIB || exit 1
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_$v.log 2>&1 || { echo "smoke FAILED for $v"; tail -3 $OUT/smoke_$v.log; exit 1; }
  if [ -n "$PYTEST_K" ]; then timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$PYTEST_K" > $OUT/pytest_$v.log 2>&1; rc=$?
  else timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_$v.log 2>&1; rc=$?; fi
  echo "parity $v rc=$rc: $(tail -1 $OUT/pytest_$v.log)"
  [ $rc -eq 0 ] || { tail -30 $OUT/pytest_$v.log; exit 1; }
done
for rep in $(seq 1 ${REPS:-2}); do for v in $VARIANTS; do
  lib=${v%%:*}; envs=""; [ "$lib" != "$v" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
  cp build/libptamd_$lib.so $LIB
  env $envs timeout -k 10 300 python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-extra $BENCH_ARGS 2>>$OUT/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['roofline']['kernel_ms_per_launch'], d['rgba_checksum_rank0_band'])" || { echo "bench FAILED for $v"; tail -5 $OUT/ab.err; exit 1; }
done; done
cp build/libptamd_default.so $LIB
