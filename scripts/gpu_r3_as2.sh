#!/bin/bash
# with LDS-typed pointers: float nodes against quantised nodes, treelet sizes, both big scenes
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
run() { env PTAMD_TUNING=1 "$@" timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 2 $ARGS 2>>$OUT/as2.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$ARGS $*', d['value'])" || exit 1; }
for ARGS in "--atrium" "--tessellate 24"; do
  for q in 0 1; do for t in 256 512 640; do run PTAMD_WIDE4Q=$q PTAMD_TREELET=$t; done; done
  run PTAMD_WIDE4Q=0 PTAMD_WALK_MIN4=16; run PTAMD_WIDE4Q=0 PTAMD_WALK_MIN4=24; run PTAMD_WIDE4Q=1 PTAMD_WALK_MIN4=16; run PTAMD_WIDE4Q=1 PTAMD_WALK_MIN4=24
done
