#!/bin/bash
# One full GPU-box session: build, GPU tests, smoke, PMC passes (-> traffic json), bench, rocprofv3 kernel trace.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash scripts/collect_pmc.sh persistent > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; tail -2 $OUT/pmc.log
cp $OUT/traffic_persistent.json $R/profiles/traffic_latest.json
cd $R
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
PTAMD_BENCH_FORCE_GATHER=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_gather.json 2>> $OUT/bench.err; echo "bench+gather rc=$?"; cut -c1-300 $OUT/bench_gather.json
for k in bvh blockwise brute; do python bench.py --steps 10 --warmup 2 --kernel $k --no-cpu-baseline > $OUT/bench_$k.json 2>> $OUT/bench.err; cut -c1-260 $OUT/bench_$k.json; echo; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kernel -- python3 $R/bench.py --steps 40 --warmup 4 --no-cpu-baseline > $OUT/prof_kernel.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof_kernel -name "*kernel_stats*" | head -2
# C++ host (examples/headless_render.cpp over raytrace.hpp) must produce the same picture as the Python host
cd $R
g++ -std=c++17 -O1 -Iinclude -Icuda-pathtracer_amd/host examples/headless_render.cpp -Lcuda-pathtracer_amd -lptamd -Wl,-rpath,$R/cuda-pathtracer_amd -o $OUT/headless_render && \
  $OUT/headless_render assets/crate_land.scene 320 180 8 $OUT/cpp.png && python - <<'PY'
import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import cuda_pathtracer_amd as P
hs = P.HostScene.load("assets/crate_land.scene")          # real textures, normal maps and the decoded cube cross on both hosts
with P.Context(0) as ctx:
    sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(P.cubemap_for_scene(hs, asset_folder="assets"))
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), 320, 180)
    fr.render(spp=8, bounces=3); torch.cuda.synchronize()
    a = fr.surface.cpu().numpy()[:, :, :3]
b = P.load_image8("gpurun_out/cpp.png")
print("C++ host == Python host:", bool(np.array_equal(a, b)))
PY
