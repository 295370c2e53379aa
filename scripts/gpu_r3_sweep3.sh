#!/bin/bash
# Builder constants after the box test got cheaper (16 VALU per box, ~67 per triangle): leaf size and the SAH's triangle cost.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
run() { env PTAMD_TUNING=1 "$@" timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 3 2>>$OUT/sweep3.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$*', d['value'], 'nodes/ray', r.get('nodes_per_ray'), 'tris/ray', r.get('tris_per_ray'), 'bvh nodes', d['config']['bvh_nodes'])" || exit 1; }
run PTAMD_BVH_MAX_LEAF=3
for ml in 1 2 4; do run PTAMD_BVH_MAX_LEAF=$ml; done
for ic in 0.5 1 1.5 2 3 4 6; do run PTAMD_BVH_ISECT_COST=$ic; done
for ic in 1 2 4; do run PTAMD_BVH_MAX_LEAF=2 PTAMD_BVH_ISECT_COST=$ic; done
run PTAMD_BVH_MAX_LEAF=3
