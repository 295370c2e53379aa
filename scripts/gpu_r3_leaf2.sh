#!/bin/bash
# leaves of at most two triangles instead of three (PTAMD_BVH_MAX_LEAF): every bench configuration, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
run() { local ml=$1; shift; PTAMD_TUNING=1 PTAMD_BVH_MAX_LEAF=$ml timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline "$@" 2>>$OUT/leaf2.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('max_leaf $ml', '$*', d['value'])" || exit 1; }
for rep in 1 2; do for ml in 3 2; do
  run $ml --steps 40
  run $ml --steps 8 --atrium
  run $ml --steps 6 --width 3840 --height 2160 --spp 16 --bounces 8 --aperture 0.113
  run $ml --steps 20 --fix-backslashes
  run $ml --steps 20 --scene assets/crate_land.scene
  run $ml --steps 8 --tessellate 24
done; done
