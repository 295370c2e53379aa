#!/bin/bash
# the first pipelined launch behind a whole-GPU one waits for it (build/libptamd_trans.so) vs not (build/libptamd_w4.so): one-stream legs, short and long
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; cd $R
LIB=cuda-pathtracer_amd/libptamd.so
for rep in 1 2 3 4; do for v in w4 trans; do
  cp build/libptamd_$v.so $LIB
  for mode in "" "--sequential"; do
    timeout -k 10 200 python bench.py --frames-in-flight 1 --no-extra --no-cpu-baseline --steps 20 --warmup 3 $mode | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', '[$mode]', '20 steps', d['value'])"
  done
done; done
cp build/libptamd_trans.so $LIB
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "back_to_back or batched or graph" 2>&1 | tail -2
