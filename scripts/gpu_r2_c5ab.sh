#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# configs[4] (3840x2160, 16 spp, 8 bounces, aperture 0.113) with and without the short reciprocal, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
for rep in 1 2; do for v in 0 1; do
  echo -n "configs[4] short_rcp=$v: "; PTAMD_SHORT_RCP=$v timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --width 3840 --height 2160 --spp 16 --bounces 8 --aperture 0.113 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d.get('value_unpipelined'), d['roofline']['kernel_ms_per_launch'])" || exit 1
done; done
