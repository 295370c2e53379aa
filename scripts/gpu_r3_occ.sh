#!/bin/bash
# How does the wide walk (atrium) scale with resident waves?  build/libptamd_<v>.so, v in $VARIANTS (o2 / o3 / w4: 2 / 3 / 4 waves per
# SIMD with the same code; o5: 5 waves at 96 VGPRs, two workgroups of 10 waves per CU; o5b: four of 5)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
for rep in 1 2; do for v in ${VARIANTS:-o2 o3 w4}; do
  cp build/libptamd_$v.so $LIB
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --atrium 2>>$OUT/occ.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['roofline']['kernel_ms_per_launch'])" || exit 1
done; done
if [ -n "$PARITY" ]; then cp build/libptamd_$PARITY.so $LIB; timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "trace_rays or wide or config4 or large_scene or fuzz or huge or atrium" 2>&1 | tail -2; fi
cp build/libptamd_w4.so $LIB
