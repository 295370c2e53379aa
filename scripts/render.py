#!/usr/bin/env python3
"""Headless render of a .scene file to a PNG or PPM through the C-ABI (what main.cpp's frame loop shows in
its window), with the scene's textures and cube cross decoded as the reference would on Linux.
usage: render.py SCENE WIDTH HEIGHT SPP OUT.(png|ppm) [BOUNCES] [POST_ID]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_pathtracer_amd as P  # noqa: E402
from cuda_pathtracer_amd.image import save_png, save_ppm  # noqa: E402

scene, w, h, spp, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
bounces = int(sys.argv[6]) if len(sys.argv) > 6 else P.REFERENCE_BOUNCES
post = int(sys.argv[7]) if len(sys.argv) > 7 else 0
hs = P.HostScene.load(scene)
with P.Context(0) as ctx:
    sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(P.cubemap_for_scene(hs, asset_folder=os.path.dirname(os.path.abspath(scene))))
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), w, h)
    fr.render(spp=spp, bounces=bounces, post_id=post, batched=True)
    torch.cuda.synchronize()
    (save_png if out.endswith(".png") else save_ppm)(out, fr.surface.cpu().numpy())
print(f"wrote {out}: {w}x{h}, {spp} spp, {bounces} bounces")
