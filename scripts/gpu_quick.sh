#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$LABEL', d['config']['kernel'], d['value'], 'Msamples/s', d['ms_per_frame'], 'ms/frame kernel', r['kernel_ms_per_launch'], 'ms', d['rgba_checksum_rank0_band'])"; }
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
LABEL=batched run --kernel persistent && LABEL=tile run --kernel bvh
