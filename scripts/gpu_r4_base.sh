#!/bin/bash
# round 4, first session: counters of the SHIPPED wide walk on the atrium (configs[3]) stamped with the build id, the instrumented
# build's lane statistics of its phases, and the bench line of the same build
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-r04_base}; mkdir -p $OUT; cd $R
timeout -k 10 180 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -5 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 300 python3 scripts/gpu_wide_stats.py > $OUT/${TAG}_wide_stats_atrium.json 2> $OUT/wide_stats.err; echo "stats rc=$?"; cat $OUT/${TAG}_wide_stats_atrium.json
PMC_OUT=pmc_atrium.json PMC_EXTRA=l1x bash scripts/collect_pmc.sh restart --atrium > $OUT/pmc_atrium.log 2>&1; echo "pmc rc=$?"; tail -3 $OUT/pmc_atrium.log
cp $OUT/pmc_summary_restart.json $OUT/${TAG}_pmc_atrium_full.json
cat $OUT/pmc_atrium.json
timeout -k 10 700 python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; cat $OUT/${TAG}_bench.json
