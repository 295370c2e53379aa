#!/bin/bash
# round 3: which part of the new restart kernel costs the steady state?  A/B of prebuilt libraries on ONE box, then the full bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
bash scripts/gpu_ab_libs.sh "$@" || exit 1
timeout -k 10 500 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/r3_bench_ab.json 2> $OUT/r3_bench_ab.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/r3_bench_ab.json"))
print(d["value"], "unpipelined", d.get("value_unpipelined"), "host_sync", d.get("value_host_sync"), "seq", d.get("value_sequential"), "seq_host_sync", d.get("value_sequential_host_sync"), [o["value"] for o in d.get("other_configs",[])])
PY
