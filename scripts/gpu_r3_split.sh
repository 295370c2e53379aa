#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# reference pre-splitting (large faces represented by several clipped references) on the atrium
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in "X=0" "PTAMD_BVH_SPLIT_ALPHA=0.01" "PTAMD_BVH_SPLIT_ALPHA=0.001" "PTAMD_BVH_SPLIT_ALPHA=0.0001 PTAMD_BVH_SPLIT_BUDGET=100000" "PTAMD_BVH_SPLIT_ALPHA=0.00002 PTAMD_BVH_SPLIT_BUDGET=200000"; do
  env $v timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 2 --atrium | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$v]', d['value'], d['roofline']['kernel_ms_per_launch'], d['roofline']['nodes_per_ray'], d['roofline']['tris_per_ray'], d['config']['bvh_nodes'])"
done
