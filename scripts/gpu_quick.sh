#!/bin/bash
# quick GPU check: build, GPU tests, bench of the three kernel variants
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
for k in persistent bvh; do
python bench.py --steps 20 --warmup 3 --kernel $k --no-cpu-baseline 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['config']['kernel'], d['value'], 'Msamples/s', d['ms_per_frame'], 'ms/frame kernel', r['kernel_ms_per_launch'], 'ms nodes/ray', r['nodes_per_ray'], 'tris/ray', r['tris_per_ray'])"
done
