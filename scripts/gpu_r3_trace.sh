#!/bin/bash
export PTAMD_TUNING=1   # the knobs below are read only with this set
# kernel start / end times of back-to-back launches on one stream (does the library's pipelining overlap them?)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cat > /tmp/b2b.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["R"])
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
import cuda_pathtracer_amd as P
R = os.environ["R"]
hs = P.HostScene.load(os.path.join(R, "assets", "indoor.scene"))
cube = P.cubemap_for_scene(hs)
frames = int(os.environ.get("FRAMES", "4"))
with P.Context(0) as ctx:
    sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(cube)
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), 1920, 1080)
    for _ in range(40):
        fr.render(spp=frames, bounces=4, kernel=P.KERNEL_BVH_RESTART, batched=True, reset=True)
    torch.cuda.synchronize()
    for use_events in (False, True):
        n = int(os.environ.get("N", "20"))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        st = torch.cuda.current_stream()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if use_events: evs[i][0].record(st)
            fr.render(spp=frames, bounces=4, kernel=P.KERNEL_BVH_RESTART, batched=True, reset=True)
            if use_events: evs[i][1].record(st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("torch events" if use_events else "no events", "ms per launch", dt / n * 1e3, "Msamples/s", 1920 * 1080 * frames * n / dt / 1e6)
PY
cd /tmp && export TMPDIR=/tmp R=$R
python3 /tmp/b2b.py
N=10 python3 /tmp/b2b.py
PTAMD_OVERLAP=0 python3 /tmp/b2b.py
cd $R
for v in "" "PTAMD_OVERLAP=0"; do env $v python3 bench.py --frames-in-flight 1 --no-extra --no-cpu-baseline --steps 10 --warmup 2 | python3 -c "import json,sys; d=json.load(sys.stdin); print('bench fif1 [$v]', d['value'], d['ms_per_step'])"; done
for v in "" "PTAMD_OVERLAP=0"; do env $v python3 bench.py --frames-in-flight 1 --no-extra --no-cpu-baseline --steps 40 --warmup 2 | python3 -c "import json,sys; d=json.load(sys.stdin); print('bench fif1 40 steps [$v]', d['value'], d['ms_per_step'])"; done
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/r3_trace -- python3 $R/bench.py --frames-in-flight 1 --no-extra --no-cpu-baseline --steps 10 --warmup 2 --settle-ms 20 > $OUT/r3_trace.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/r3_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "pt_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-70]["Start_Timestamp"])
for r in rows[-70:]:
    print(r["Kernel_Name"][:40].ljust(40), r.get("Queue_Id"), r.get("Stream_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Grid_Size"))
PY
