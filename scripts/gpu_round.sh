#!/bin/bash
# One GPU-box session: build, GPU tests, smoke, bench, rocprofv3 kernel trace.  Run via gpurun.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
python bench.py --steps 10 --warmup 2 --kernel brute --no-cpu-baseline > $OUT/bench_brute.json 2>> $OUT/bench.err; cat $OUT/bench_brute.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kernel -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_kernel.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof_kernel -name "*stats*" | head
