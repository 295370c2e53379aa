#!/bin/bash
# quantised four-wide walk (PTAMD_WIDE4Q=1) with a fifth wave per SIMD: build/libptamd_q.so (one 16-wave workgroup per CU, 4 per SIMD),
# _q4w.so (four workgroups of 4 waves, path state parked in LDS), _q5w.so (five of them: 5 per SIMD)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
for rep in 1 2; do for v in ${VARIANTS:-q q4w q5w}; do
  cp build/libptamd_$v.so $LIB
  for args in "--atrium" "--tessellate 24"; do
    PTAMD_TUNING=1 PTAMD_WIDE4Q=1 timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra $args 2>>$OUT/q5.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', '$args', d['value'], d['roofline']['kernel_ms_per_launch'])" || exit 1
  done
done; done
if [ -n "$PARITY" ]; then cp build/libptamd_$PARITY.so $LIB; PTAMD_TUNING=1 PTAMD_WIDE4Q=1 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "quantised_four or trace_rays or wide or config4 or large_scene or fuzz or huge or atrium" 2>&1 | tail -2; fi
cp build/libptamd_q.so $LIB
