#!/bin/bash
# round 4 full session: smoke, GPU suite, PMC passes of EVERY bench configuration (one record per workload, stamped with the build id:
# profiles/pmc_latest.json = headline, pmc_atrium.json, pmc_c4k.json, pmc_textured_indoor.json, pmc_crate_land.json), the bench line
# (roofline on the headline and on every other_configs entry), rocprofv3 kernel stats of the headline launches.
#   gpu_round4.sh [tag]        SKIP_PMC=1: reuse the records already in profiles/       SKIP_TESTS=1
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-r04_v1}; mkdir -p $OUT; cd $R
timeout -k 10 180 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -5 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/${TAG}_pytest_gpu.log
  [ $rc -eq 0 ] || exit 1
fi
if [ -z "$SKIP_PMC" ]; then
  pmc() {   # name, record file, bench args...
    local name=$1 rec=$2; shift 2
    PMC_OUT=$rec PMC_EXTRA=l1x bash scripts/collect_pmc.sh restart "$@" > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$? $(grep -c 'rc=0' $OUT/pmc_$name.log) passes ok"
    cp $OUT/pmc_summary_restart.json $OUT/${TAG}_pmc_${name}_full.json; cp $OUT/$rec $R/profiles/$rec; cp $OUT/$rec $OUT/${TAG}_$rec
  }
  pmc headline pmc_latest.json
  pmc atrium pmc_atrium.json --atrium
  pmc c4k pmc_c4k.json --width 3840 --height 2160 --spp 16 --bounces 8 --aperture 0.113
  pmc textured_indoor pmc_textured_indoor.json --fix-backslashes
  pmc crate_land pmc_crate_land.json --scene $R/assets/crate_land.scene
fi
cd $R
timeout -k 10 900 python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; cat $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kernel -- python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extra > $OUT/prof_kernel.log 2>&1; echo "rocprof rc=$?"
cp $(find $OUT/prof_kernel -name "*kernel_stats*" | head -1) $OUT/${TAG}_kernel_stats.csv
head -6 $OUT/${TAG}_kernel_stats.csv
