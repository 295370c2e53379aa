#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$LABEL', d['config']['kernel'], d['value'], 'Msamples/s', 'kernel', r['kernel_ms_per_launch'], 'ms', d['rgba_checksum_rank0_band'])"; }
python -c "import __graft_entry__ as g; g.build()" || exit 1
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for k in bvh persistent blockwise; do LABEL=default run --kernel $k; done
for cfg in "1024 8" "512 8" "512 5" "256 6"; do set -- $cfg
  rm -f cuda-pathtracer_amd/libptamd.so; make -s lib EXTRA_HIPFLAGS="-DPT_TILE_THREADS=$1 -DPT_TILE_WAVES_PER_EU=$2" 2>&1 | grep -E "error"
  LABEL="tile$1-w$2" run --kernel bvh
done
rm -f cuda-pathtracer_amd/libptamd.so; make -s lib 2>&1 | grep error; true
