#!/bin/bash
# what PC-sampling configurations does this box offer?
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1; echo "rc=$?"
grep -n -i -B2 -A12 "pc.sampl" $OUT/list_avail.txt | head -60
