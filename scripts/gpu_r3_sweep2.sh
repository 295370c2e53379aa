#!/bin/bash
# After the box loop got cheaper (17 VALU per iteration): do the phase thresholds want new values?  Headline, 40 steps each.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
run() { env PTAMD_TUNING=1 "$@" timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 3 2>>$OUT/sweep2.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$*', d['value'])" || exit 1; }
run PTAMD_WALK_MIN=7
for w in 4 5 6 8 9 10 12; do run PTAMD_WALK_MIN=$w; done
run PTAMD_WALK_MIN=7
for rm in 8 12 20 24 32; do run PTAMD_ROUND_MIN=$rm; done
for rd in 2 3 5 6 8; do run PTAMD_ROUND_DIV=$rd; done
run PTAMD_WALK_MIN=7
