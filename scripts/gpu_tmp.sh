R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
bash scripts/gpu_round.sh || exit 1
cp $OUT/pmc_summary_restart.json $OUT/pmc_summary_restart_indoor.json
cp $OUT/pmc_latest.json $OUT/pmc_latest_indoor.json
rm -rf $OUT/pmc
PMC_EXTRA=l1x bash scripts/collect_pmc.sh restart --atrium > $OUT/pmc_atrium.log 2>&1; echo "pmc atrium rc=$?"; tail -1 $OUT/pmc_atrium.log
cp $OUT/pmc_summary_restart.json $OUT/pmc_summary_restart_atrium_wide.json
cp $OUT/pmc_latest_indoor.json $OUT/pmc_latest.json
cp $OUT/pmc_summary_restart_indoor.json $OUT/pmc_summary_restart.json
