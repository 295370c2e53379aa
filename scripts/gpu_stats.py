#!/usr/bin/env python3
"""Instrumented-build traversal statistics of the headline frame (one 1-spp launch): wave-level loop trip counts and lane utilisation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_pathtracer_amd as P  # noqa: E402

W, H, B = 1920, int(os.environ.get("HEIGHT", 1080)), int(os.environ.get("BOUNCES", 4))
hs = P.HostScene.load(os.path.join(ROOT, "assets", "indoor.scene"))
with P.Context(0) as ctx:
    sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(P.cubemap_for_scene(hs))
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H)
    for name, k in (("tile", P.KERNEL_BVH), ("persistent", P.KERNEL_BVH_PERSISTENT)):
        fr.reset()
        l = ctx.make_launch(fr.surface, fr.accum, sid, cid, hs.camera_struct(), W, H, frame_nb=1, bounces=B, kernel=k)
        s = ctx.raytrace_stats(l)
        print(name, {k2: v for k2, v in s.items() if v})
        print("  box-loop lane utilisation %.3f, tri-loop lane utilisation %.3f, box iters/ray(wave) %.2f" % (
            s["nodes_visited"] / (64.0 * max(s["wave_node_iters"], 1)), s["tris_tested"] / (64.0 * max(s["wave_tri_iters"], 1)),
            s["wave_node_iters"] * 64.0 / max(s["rays"], 1)))
