#!/bin/bash
# Stability soak: long benches of every scheduling path, the GPU suite three times over, under hard time limits.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 90 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; exit 1; }
for i in 1 2 3; do
  timeout -k 10 300 python -m pytest tests -m gpu -x -q -p timeout --timeout 120 --timeout-method thread > $OUT/pytest_soak_$i.log 2>&1 || { echo "pytest pass $i FAILED"; tail -5 $OUT/pytest_soak_$i.log; exit 1; }
  echo "pytest pass $i: $(tail -1 $OUT/pytest_soak_$i.log)"
done
for args in "--steps 400" "--steps 300 --frames-in-flight 3" "--steps 200 --as-rank 3/8 --frames-in-flight 3" "--steps 200 --as-rank 3/8 --interleave 8 --frames-in-flight 3" "--steps 100 --bounces 8" "--steps 60 --width 3840 --height 2160 --spp 16 --bounces 8 --aperture 0.113" "--steps 100 --kernel persistent" "--steps 100 --kernel split" "--steps 100 --width 1001 --height 333" "--steps 40 --atrium" "--steps 60 --sequential" "--steps 300 --frames-in-flight 1" "--steps 200 --frames-in-flight 1 --sequential" "--steps 100 --fix-backslashes" "--steps 100 --scene assets/crate_land.scene"; do
  timeout -k 10 200 python bench.py --warmup 3 --no-cpu-baseline --no-extra $args 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args ->', d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step, checksum', d['rgba_checksum_rank0_band'])" || { echo "bench $args FAILED or hung"; exit 1; }
done
echo soak OK
