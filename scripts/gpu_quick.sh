#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$LABEL', d['value'], 'Msamples/s', d['ms_per_frame'], 'ms/frame kernel', r['kernel_ms_per_launch'], 'ms', d['config']['frames_in_flight'], d['rgba_checksum_rank0_band'])"; }
python -c "import __graft_entry__ as g; g.build()" || exit 1
for f in 1 2 3; do for h in 135 270 1080; do LABEL="rows$h-inflight$f" PTAMD_BENCH_FORCE_GATHER=1 run --kernel persistent --height $h --frames-in-flight $f; done; done
tail -3 $OUT/bench.err
