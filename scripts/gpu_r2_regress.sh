#!/bin/bash
# VALU instructions per launch (PMC) next to the wave-level loop counts (STATS build) for 1..4 bounces at 4 spp:
# the data for "where do the instructions go" (box iterations x 21, triangle iterations, per-round and per-sample work)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/regress; rm -rf $OUT; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
for B in 1 2 3 4; do
  echo "== bounces $B"
  ONLY=restart SPP=4 BOUNCES=$B timeout -k 10 120 python scripts/gpu_stats.py 2>/dev/null | tee $OUT/stats_b$B.txt || exit 1
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc_b$B -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --bounces $B > $OUT/pmc_b$B.log 2>&1 ) || { echo "pmc failed"; tail -3 $OUT/pmc_b$B.log; exit 1; }
  python3 - $OUT/pmc_b$B <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if "restart<true, false>" in r["Kernel_Name"]:      # the un-instrumented batched dispatches (not the STATS launches of the bench)
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in acc.items():
    print("  %s per launch: %.0f (%d dispatches)" % (c, sum(d.values()) / len(d), len(d)))
PY
done
