#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# knob sweeps on configs[4] (4K, 16 spp, 8 bounces, aperture 0.113) and on the atrium's leaf size, one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
run() { env $1 timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --steps ${STEPS:-6} --warmup 2 $2 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('[$1]', '$2', d['value'], d['roofline']['kernel_ms_per_launch'])"; }
C4="--width 3840 --height 2160 --spp 16 --bounces 8 --aperture 0.113"
for v in "X=0" "PTAMD_ROUND_MIN=24" "PTAMD_ROUND_MIN=32" "PTAMD_ROUND_MIN=8" "PTAMD_ROUND_DIV=3" "PTAMD_ROUND_DIV=6" "PTAMD_WALK_MIN=5" "PTAMD_WALK_MIN=10" "PTAMD_ROUND_MIN=24 PTAMD_ROUND_DIV=3"; do STEPS=4 run "$v" "$C4"; done
for v in "X=0" "PTAMD_BVH_MAX_LEAF=4" "PTAMD_BVH_MAX_LEAF=2" "PTAMD_BVH_ISECT_COST=1.0" "PTAMD_BVH_ISECT_COST=3.0" "PTAMD_TREELET=640" "PTAMD_TREELET=384"; do STEPS=8 run "$v" "--atrium"; done
