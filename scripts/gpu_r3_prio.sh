#!/bin/bash
# internal streams at lowest priority vs default: one-stream back-to-back legs of the bench, several repetitions (short timed regions: the
# transition from a whole-GPU launch to pipelined half-GPU launches is inside them)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
for rep in 1 2 3; do for v in w4 prio; do
  cp build/libptamd_$v.so $LIB
  for mode in "" "--sequential"; do
    timeout -k 10 200 python bench.py --frames-in-flight 1 --no-extra --no-cpu-baseline --steps 10 --warmup 2 $mode | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', '[$mode]', '10 steps', d['value'])"
    timeout -k 10 200 python bench.py --frames-in-flight 1 --no-extra --no-cpu-baseline --steps 40 --warmup 2 $mode | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', '[$mode]', '40 steps', d['value'])"
  done
done; done
cp build/libptamd_prio.so $LIB
