#!/bin/bash
# Is the 8-rank proxy bound by the host loop (launch + resolve + all-gather per 0.125 ms frame)?  Same rank share with almost no
# path work (1 spp, 1 bounce), with and without the forced all-gather.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
one() { # label, env gather, args...
  local label=$1 g=$2; shift 2
  PTAMD_BENCH_FORCE_GATHER=$g timeout -k 10 200 python bench.py --as-rank 3/8 --interleave 8 --no-extra --no-cpu-baseline --steps 200 --warmup 5 "$@" \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$label', 'ms/frame', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_ms_per_launch'])"
}
one "full work, gather, 4 in flight   " 1 --frames-in-flight 4 || exit 1
one "full work, no gather, 4 in flight" 0 --frames-in-flight 4 || exit 1
one "tiny work, gather, 4 in flight   " 1 --frames-in-flight 4 --spp 1 --bounces 1 || exit 1
one "tiny work, no gather, 4 in flight" 0 --frames-in-flight 4 --spp 1 --bounces 1 || exit 1
one "tiny work, gather, 1 in flight   " 1 --frames-in-flight 1 --spp 1 --bounces 1 || exit 1
one "tiny work, no gather, 1 in flight" 0 --frames-in-flight 1 --spp 1 --bounces 1 || exit 1
one "full work, gather, 8 in flight/8 " 1 --frames-in-flight 8 --machine-share 8 || exit 1
one "full work, no gather, 8 in flight/8" 0 --frames-in-flight 8 --machine-share 8 || exit 1
