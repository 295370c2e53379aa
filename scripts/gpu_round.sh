#!/bin/bash
# One full GPU-box session: build check, GPU tests, smoke, PMC passes (-> pmc_latest.json), bench, rocprofv3 kernel trace.
#   bash scripts/gpu_round.sh [tag]      artefacts land in gpurun_out/; copy the ones to keep into profiles/<tag>_*
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
KERNEL=${KERNEL:-restart}
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "smoke FAILED or hung"; tail -5 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
bash scripts/collect_pmc.sh $KERNEL > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; tail -2 $OUT/pmc.log
cp $OUT/pmc_latest.json $R/profiles/pmc_latest.json
cd $R
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --kernel $KERNEL > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
# the same command, headline launches only, so that the per-kernel average is the duration of a headline launch
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kernel -- python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extra --kernel $KERNEL > $OUT/prof_kernel.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof_kernel -name "*kernel_stats*" | head -2
cat $(find $OUT/prof_kernel -name "*kernel_stats*" | head -1) | head -8
tail -1 $OUT/prof_kernel.log | cut -c1-400
