#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# round 3: shared pools + launch overlap.  smoke, GPU suite, launch timelines with / without pool sharing, bench with / without overlap
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 180 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r3_smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -5 $OUT/r3_smoke.log; exit 1; }
tail -1 $OUT/r3_smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/r3_pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/r3_pytest_gpu.log
[ $rc -eq 0 ] || exit 1
PTAMD_TAIL=0 timeout -k 10 200 python scripts/gpu_timeline.py --out $OUT/r3_tail_off.json || exit 1
timeout -k 10 200 python scripts/gpu_timeline.py --out $OUT/r3_tail_on.json || exit 1
timeout -k 10 200 python scripts/gpu_timeline.py --frames 1 --share 4 --interleave 8:3:8 --out $OUT/r3_tail_on_rank3of8.json || exit 1
PTAMD_TAIL=0 timeout -k 10 200 python scripts/gpu_timeline.py --frames 1 --share 4 --interleave 8:3:8 --out $OUT/r3_tail_off_rank3of8.json || exit 1
for v in "" "PTAMD_OVERLAP=0" "PTAMD_TAIL=0" "PTAMD_OVERLAP=0 PTAMD_TAIL=0"; do
  tag=$(echo "$v" | tr ' =' '__'); [ -z "$tag" ] && tag=default
  env $v timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/r3_bench_$tag.json 2> $OUT/r3_bench_$tag.err; echo "bench [$v] rc=$?"
  python - <<PY
import json
d=json.load(open("$OUT/r3_bench_$tag.json"))
print("$tag", d["value"], d.get("value_unpipelined"), d.get("value_host_sync"), d.get("value_sequential"), d.get("value_sequential_host_sync"), [o["value"] for o in d.get("other_configs",[])])
PY
done
