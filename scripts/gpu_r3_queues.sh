#!/bin/bash
# HIP maps streams onto 4 hardware queues by default: 8 frames in flight then serialise pairwise.  GPU_MAX_HW_QUEUES=8/16.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
one() { # label, queues, args...
  local label=$1 q=$2; shift 2
  GPU_MAX_HW_QUEUES=$q PTAMD_BENCH_FORCE_GATHER=1 timeout -k 10 200 python bench.py --as-rank 3/8 --interleave 8 --no-extra --no-cpu-baseline --steps 200 --warmup 8 "$@" \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$label queues=$q', 'ms/frame', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_ms_per_launch'])"
}
for q in 4 8 16; do
  one "4 in flight on quarters" $q --frames-in-flight 4 || exit 1
  one "8 in flight on eighths " $q --frames-in-flight 8 --machine-share 8 || exit 1
  one "8 in flight on quarters" $q --frames-in-flight 8 --machine-share 4 || exit 1
  one "6 in flight on quarters" $q --frames-in-flight 6 --machine-share 4 || exit 1
done
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 40 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('headline with 8 queues', d['value'])"
