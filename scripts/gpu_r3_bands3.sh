#!/bin/bash
# 8-rank proxy with more, smaller launches in flight: 6 on sixths, 8 on eighths, 8 on sixths (the tail of a launch is a fixed ~150 us)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
for cfg in "4 4" "6 6" "8 8" "8 6" "12 8"; do set -- $cfg
  timeout -k 10 400 python scripts/band_proxy.py --ranks 8 --in-flight $1 --machine-share $2 --interleave 8 --steps 96 --out $OUT/r3_bands3_$1_$2.json || exit 1
done
