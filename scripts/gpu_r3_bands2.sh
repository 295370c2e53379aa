#!/bin/bash
# more frames in flight than GPU shares: do the queued workgroups of one launch fill the others' tails?  rank 3 of 8, interleaved 8-row bands
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
for cfg in "4 4" "5 4" "6 4" "8 4" "6 3" "8 6" "6 6" "3 2" "4 2"; do set -- $cfg
  PTAMD_BENCH_FORCE_GATHER=1 timeout -k 10 200 python bench.py --as-rank 3/8 --interleave 8 --steps 60 --warmup 5 --no-cpu-baseline --no-extra --frames-in-flight $1 --machine-share $2 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('in flight $1 share 1/$2:', d['ms_per_step'], 'ms per frame ->', round(1920*1080*4/d['ms_per_step']/1e3,1), 'Msamples/s implied at 8 ranks')"
done
