#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
run() { label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra --kernel restart $BARGS 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], r['kernel_ms_per_launch'], r['nodes_per_ray'], r['tris_per_ray'])"
}
BARGS="--tessellate 24" run "tessellated indoor wide wm4=16" PTAMD_WALK_MIN4=16
BARGS="--tessellate 24 --kernel persistent" run "tessellated indoor binary" X=1
PMC_EXTRA=l1x bash scripts/collect_pmc.sh restart --atrium > $OUT/pmc_atrium.log 2>&1; echo "pmc rc=$?"; tail -2 $OUT/pmc_atrium.log
cp $OUT/pmc_summary_restart.json $OUT/pmc_summary_restart_atrium_wide.json
for v in "w5=-DPT_RS4_WAVES_PER_EU=5" "w6=-DPT_RS4_WAVES_PER_EU=6" "w3=-DPT_RS4_WAVES_PER_EU=3"; do
  name=${v%%=*}; flags=${v#*=}
  make -s -B lib EXTRA_HIPFLAGS="$flags" 2>>$OUT/flags.err || { echo "$name: build failed"; continue; }
  BARGS="--atrium" run "$name atrium wide wm4=16" PTAMD_WALK_MIN4=16
done
make -s -B lib 2>>$OUT/flags.err
