#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# VALU / LDS instruction counts of the headline launch with the triangle phase spread on and off (one rocprofv3 --pmc pass each)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
cp $R/build/libptamd_spread.so $R/cuda-pathtracer_amd/libptamd.so
for v in 1 0; do
  rm -rf $OUT/pmc_spread$v
  PTAMD_LEAF_SPREAD=$v rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $OUT/pmc_spread$v -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_spread$v.log 2>&1; echo "pass spread=$v rc=$?"
  python3 - $OUT/pmc_spread$v $v <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pt_megakernel_restart<true,0>" not in row["Kernel_Name"].replace(" ", ""): continue
        a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
r = {k: v[0] / v[1] for k, v in acc.items()}
print("spread", sys.argv[2], {k: round(v / 1e6, 2) for k, v in r.items()}, "VALU per sample", round(r["SQ_INSTS_VALU"] / 8294400, 2),
      "active lanes", round(r["SQ_THREAD_CYCLES_VALU"] / r["SQ_ACTIVE_INST_VALU"], 2))
PY
done
