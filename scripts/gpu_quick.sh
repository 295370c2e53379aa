#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import cuda_pathtracer_amd as P
A = os.path.join(os.getcwd(), "assets")
hs = P.HostScene.load(A + "/crate_land.scene", image_loader=P.pil_image_loader)
cube = P.cubemap_for_scene(hs, asset_folder=A, image_loader=P.pil_image_loader)
with P.Context(0) as ctx:
    sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(cube)
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), 1920, 1080)
    fr.render(spp=4, bounces=4, batched=True); torch.cuda.synchronize()
    t = time.time()
    for _ in range(10):
        fr.accum.zero_(); fr.render(spp=4, bounces=4, batched=True)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print(f"crate_land textured 1080p 4spp B4: {dt*1e3:.3f} ms/frame {1920*1080*4/dt/1e6:.0f} Msamples/s")
    fr2 = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), 640, 360)
    fr2.render(spp=128, bounces=3, batched=True); torch.cuda.synchronize()
    P.save_ppm("gpurun_out/crate_land.ppm", fr2.surface.cpu().numpy())
PY
