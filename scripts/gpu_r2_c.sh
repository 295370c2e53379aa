#!/bin/bash
# band proxy + PMC of the L2-resident walk on the atrium (configs[3])
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout -k 10 900 python scripts/band_proxy.py --out $OUT/band_proxy.json; echo "proxy rc=$?"
PMC_EXTRA=l1 bash scripts/collect_pmc.sh restart --atrium > $OUT/pmc_atrium.log 2>&1; echo "pmc rc=$?"; tail -3 $OUT/pmc_atrium.log
cp $OUT/pmc_summary_restart.json $OUT/pmc_summary_restart_atrium.json
cat $OUT/pmc_summary_restart_atrium.json
