#!/bin/bash
# the eight-wide quantised walk once more, now that the walk's LDS accesses are LDS instructions (it had 16 FLAT ones)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
run() { env PTAMD_TUNING=1 "$@" timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --steps 8 --warmup 2 $ARGS 2>>$OUT/w8again.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$ARGS $*', d['value'])" || exit 1; }
for ARGS in "--atrium" "--tessellate 24"; do for rep in 1 2; do run PTAMD_WIDE8=0; run PTAMD_WIDE8=1; run PTAMD_WIDE4Q=1; done; done
