"""First GPU bring-up: parity vs oracle at several sizes + quick timing of both kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cuda_pathtracer_amd as P, pt_oracle as O

hs = P.HostScene.load(os.path.join(ROOT, "assets", "indoor.scene"))
cube = P.cubemap_for_scene(hs)
osc = O.OracleScene.from_host_scene(hs, cube)
ocam = O.camera_from_record(hs.camera)
ctx = P.Context(0)
sid = ctx.upload_scene(hs); cid = ctx.upload_cubemap(cube)
print("scene info", ctx.scene_info(sid), flush=True)
for (W, H, spp, B) in [(64, 64, 2, 3), (256, 256, 1, 2), (250, 130, 2, 4)]:
    ref_acc, ref_rgba = O.render(osc, ocam, W, H, spp=spp, bounces=B)
    for kern in (P.KERNEL_BRUTE_FORCE, P.KERNEL_BVH):
        fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H)
        fr.render(spp=spp, bounces=B, kernel=kern)
        torch.cuda.synchronize()
        acc = fr.accum.cpu().numpy(); rgba = fr.surface.cpu().numpy()
        nbad = int((acc.view(np.uint32) != ref_acc.view(np.uint32)).any(axis=2).sum())
        nbad8 = int((rgba != ref_rgba).any(axis=2).sum())
        maxd = float(np.abs(acc - ref_acc).max())
        print(f"{W}x{H} spp{spp} B{B} kernel{kern}: accum-mismatch px {nbad}, rgba-mismatch px {nbad8}, max|d| {maxd:.3g}", flush=True)
# timing at 1080p
W, H, spp, B = 1920, 1080, 4, 4
for kern in (P.KERNEL_BRUTE_FORCE, P.KERNEL_BVH):
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H)
    fr.render(spp=spp, bounces=B, kernel=kern); torch.cuda.synchronize()
    a1 = fr.accum.clone()
    ts = []
    for it in range(3):
        fr.reset(); torch.cuda.synchronize(); t = time.time()
        fr.render(spp=spp, bounces=B, kernel=kern); torch.cuda.synchronize(); ts.append(time.time() - t)
    print(f"1080p spp4 B4 kernel{kern}: {min(ts)*1e3:.3f} ms/frame  {W*H*spp/min(ts)/1e6:.1f} Msamples/s", flush=True)
    if kern == P.KERNEL_BRUTE_FORCE: brute = a1
    else:
        nb = int((a1.view(torch.int32) != brute.view(torch.int32)).any(dim=2).sum())
        print("1080p BVH vs brute-force accumulators: mismatching pixels", nb, flush=True)
l = ctx.make_launch(fr.surface, fr.accum, sid, cid, hs.camera_struct(), W, H, frame_nb=1, bounces=4, kernel=P.KERNEL_BVH)
print("stats bvh", ctx.raytrace_stats(l))
l.kernel = P.KERNEL_BRUTE_FORCE
print("stats brute", ctx.raytrace_stats(l))
