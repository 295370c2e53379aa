#!/bin/bash
# configs[3] after round 3's work: parity + PMC of the four-wide walk (shipped) and of the eight-wide quantised nodes (PT_WIDE8=1 build)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
VARIANTS="w4 w4:PTAMD_WIDE8=1" PARITY_LIB=w4 bash scripts/gpu_r3_wide.sh || exit 1
bash scripts/gpu_r3_wide_pmc.sh w8 || exit 1
bash scripts/gpu_r3_wide_pmc.sh w4 || exit 1
cp build/libptamd_w4.so cuda-pathtracer_amd/libptamd.so
