#!/bin/bash
# quantised four-wide walk: treelet size (PTAMD_TREELET counts 128-byte units: twice as many 64-byte nodes), walk_min4, stack entries
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
run() { env PTAMD_TUNING=1 PTAMD_WIDE4Q=1 "$@" timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --atrium 2>>$OUT/q4b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$*', d['value'])" || exit 1; }
run PTAMD_TREELET=512
for t in 0 128 256 384 640 704; do run PTAMD_TREELET=$t; done
for w in 16 24 40 48; do run PTAMD_WALK_MIN4=$w; done
run PTAMD_POOL_LDS=0
run PTAMD_POOL_LDS=0 PTAMD_TREELET=768
run PTAMD_TREELET=512
