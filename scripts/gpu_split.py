"""Bring-up of the split (shader/traverser) kernel: parity vs tile kernel, error count, timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np, torch
import cuda_pathtracer_amd as P
hs = P.HostScene.load(os.path.join(ROOT, "assets", "indoor.scene"))
cube = P.cubemap_for_scene(hs)
ctx = P.Context(0)
sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(cube)
for (W, H, spp, B) in [(64, 64, 1, 1), (64, 64, 2, 3), (250, 130, 2, 4), (1920, 1080, 1, 4)]:
    out = {}
    for name, k in (("tile", P.KERNEL_BVH), ("split", P.KERNEL_BVH_SPLIT)):
        fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H)
        fr.render(spp=spp, bounces=B, kernel=k); torch.cuda.synchronize()
        out[name] = (fr.accum.cpu().numpy(), fr.surface.cpu().numpy())
    bad = int((out["tile"][0].view(np.uint32) != out["split"][0].view(np.uint32)).any(axis=2).sum())
    print(f"{W}x{H} spp{spp} B{B}: mismatching pixels {bad}, protocol time-outs {ctx.device_error_count()}", flush=True)
W, H, spp, B = 1920, 1080, 4, 4
for name, k in (("persistent", P.KERNEL_BVH_PERSISTENT), ("split", P.KERNEL_BVH_SPLIT)):
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H)
    fr.render(spp=spp, bounces=B, kernel=k, batched=True); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        fr.accum.zero_(); torch.cuda.synchronize(); t = time.time()
        fr.render(spp=spp, bounces=B, kernel=k, batched=True); torch.cuda.synchronize(); ts.append(time.time() - t)
    print(f"{name}: {min(ts)*1e3:.3f} ms/frame {W*H*spp/min(ts)/1e6:.0f} Msamples/s  time-outs {ctx.device_error_count()}", flush=True)
l = ctx.make_launch(fr.surface, fr.accum, sid, cid, hs.camera_struct(), W, H, frame_nb=1, bounces=4, kernel=P.KERNEL_BVH_SPLIT)
s = ctx.raytrace_stats(l); print("split stats", s, "node util %.3f" % (s["nodes_visited"] / max(64 * s["wave_node_iters"], 1)))
