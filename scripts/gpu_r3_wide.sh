#!/bin/bash
# round 3: eight-wide quantised nodes against round 2's four-wide ones (prebuilt libraries build/libptamd_w8.so / _w4.so): parity of
# the wide walk, then the atrium and the tessellated indoor at 1080p x 4 spp
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
cp build/libptamd_${PARITY_LIB:-w8}.so $LIB
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED"; tail -3 $OUT/smoke.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "trace_rays or wide or config4 or large_scene or fuzz or huge" > $OUT/r3_pytest_wide.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/r3_pytest_wide.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do for v in ${VARIANTS:-w4 w8}; do
  lib=${v%%:*}; envs=""; [ "$lib" != "$v" ] && envs=$(echo "${v#*:}" | tr ',' ' ')     # "lib:ENV=val,ENV=val"
  cp build/libptamd_$lib.so $LIB
  for args in "--atrium"; do
    env $envs timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra $args $BENCH_ARGS 2>>$OUT/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', '$args', d['value'], d['roofline']['kernel_ms_per_launch'], d['roofline'].get('nodes_per_ray'), d['roofline'].get('tris_per_ray'))" || exit 1
  done
done; done
cp build/libptamd_${PARITY_LIB:-w8}.so $LIB
