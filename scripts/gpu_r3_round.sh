#!/bin/bash
# round 3 full session: GPU suite, PMC passes of the headline (-> profiles/pmc_latest.json, stamped with the build id) and of the
# two textured scenes (vector-L1 / L2 counters included), bench line, rocprofv3 kernel stats of the headline launches
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r03_v1}
mkdir -p $OUT
cd $R
timeout -k 10 180 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -5 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/${TAG}_pytest_gpu.log
[ $rc -eq 0 ] || exit 1
if [ -z "$SKIP_PMC" ]; then
bash scripts/collect_pmc.sh restart > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; tail -2 $OUT/pmc.log
cp $OUT/pmc_latest.json $R/profiles/pmc_latest.json; cp $OUT/pmc_summary_restart.json $OUT/${TAG}_pmc_restart.json
PMC_EXTRA=l1x bash scripts/collect_pmc.sh restart --fix-backslashes > $OUT/pmc_tex.log 2>&1; echo "pmc textured indoor rc=$?"
cp $OUT/pmc_summary_restart.json $OUT/${TAG}_pmc_textured_indoor.json
PMC_EXTRA=l1x bash scripts/collect_pmc.sh restart --scene $R/assets/crate_land.scene > $OUT/pmc_crate.log 2>&1; echo "pmc crate_land rc=$?"
cp $OUT/pmc_summary_restart.json $OUT/${TAG}_pmc_textured_crate_land.json
cp $R/profiles/pmc_latest.json $OUT/pmc_latest.json
fi
cd $R
timeout -k 10 700 python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; cat $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kernel -- python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extra > $OUT/prof_kernel.log 2>&1; echo "rocprof rc=$?"
cp $(find $OUT/prof_kernel -name "*kernel_stats*" | head -1) $OUT/${TAG}_kernel_stats.csv
head -6 $OUT/${TAG}_kernel_stats.csv
