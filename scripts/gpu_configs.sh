#!/bin/bash
# BASELINE.json configs[3] and configs[4] on one GPU
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
python -m pytest tests -m gpu -x -q -k "config4 or config1" > $OUT/pytest_cfg.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_cfg.log
for k in persistent bvh; do
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --tessellate 24 --kernel $k > $OUT/bench_c4_$k.json 2>>$OUT/bench.err; cut -c1-120 $OUT/bench_c4_$k.json; python -c "
import json; d=json.load(open('$OUT/bench_c4_$k.json')); print(d['config']['workload'], d['config']['kernel'], d['value'], 'Msamples/s', d['roofline'])"
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --width 3840 --height 2160 --spp 16 --bounces 8 --aperture 0.113 > $OUT/bench_c5.json 2>>$OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_c5.json')); print(d['config']['workload'], d['value'], 'Msamples/s', d['ms_per_frame'], 'ms/frame', d['roofline']['kernel_ms_per_launch'])"
