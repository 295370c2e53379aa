#!/usr/bin/env python3
"""Instrumented-build traversal statistics of the headline frame (one 1-spp launch): wave-level loop trip counts and lane utilisation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_pathtracer_amd as P  # noqa: E402

W, H, B = 1920, int(os.environ.get("HEIGHT", 1080)), int(os.environ.get("BOUNCES", 4))
SPP = int(os.environ.get("SPP", 1))            # frames per launch (the batched launch of the bench is 4)
ONLY = os.environ.get("ONLY")                  # e.g. ONLY=restart
hs = P.HostScene.load(os.path.join(ROOT, "assets", "indoor.scene"))
with P.Context(0) as ctx:
    sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(P.cubemap_for_scene(hs))
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H)
    for name, k in (("tile", P.KERNEL_BVH), ("persistent", P.KERNEL_BVH_PERSISTENT), ("restart", P.KERNEL_BVH_RESTART)):
        if (ONLY and name != ONLY) or (SPP > 1 and name == "tile"):
            continue
        fr.reset()
        l = ctx.make_launch(fr.surface, fr.accum, sid, cid, hs.camera_struct(), W, H, frame_nb=1, bounces=B, kernel=k,
                            frame_count=SPP)
        s = ctx.raytrace_stats(l)
        print(name, {k2: v for k2, v in s.items() if v})
        print("  box-loop lane utilisation %.3f, tri-loop lane utilisation %.3f, box iters/ray(wave) %.2f" % (
            s["nodes_visited"] / (64.0 * max(s["wave_node_iters"], 1)), s["tris_tested"] / (64.0 * max(s["wave_tri_iters"], 1)),
            s["wave_node_iters"] * 64.0 / max(s["rays"], 1)))
        slots = 64.0 * max(s["wave_node_iters"], 1)
        print("  box-loop lane slots: active %.3f, no ray in this call %.3f, walk over %.3f, parked at a leaf %.3f" % (
            s["nodes_visited"] / slots, s["idle_unstarted"] / slots, s["idle_finished"] / slots, s["idle_parked"] / slots))
        print("  per sample: box iterations (wave) %.3f, triangle iterations (wave) %.3f" % (
            s["wave_node_iters"] / s["samples"], s["wave_tri_iters"] / s["samples"]))
        if name == "restart":
            print("  rounds per wave-tile %.2f, walks completed per round %.1f" % (
                s["fetch_events"] * 64.0 / s["samples"], s["fetch_rays"] / max(s["fetch_events"], 1)))
