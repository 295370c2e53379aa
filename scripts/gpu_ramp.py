#!/usr/bin/env python3
"""Per-launch duration of the headline launch from a cold start: how long the chip takes to settle (clock ramp / power
management) — the reason `bench.py --steps 20` reads lower than `--steps 400` on the same box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_pathtracer_amd as P  # noqa: E402

W, H, SPP, B, N = 1920, 1080, 4, 4, int(os.environ.get("N", 400))
hs = P.HostScene.load(os.path.join(ROOT, "assets", "indoor.scene"))
with P.Context(0) as ctx:
    sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(P.cubemap_for_scene(hs))
    fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H)
    idle = float(os.environ.get("IDLE", 0))
    for rep in range(2):
        if idle:
            time.sleep(idle)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
        evs[0].record()
        for i in range(N):
            fr.render(spp=SPP, bounces=B, kernel=P.KERNEL_BVH_RESTART, batched=True, reset=True)
            evs[i + 1].record()
        torch.cuda.synchronize()
        ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
        pick = [0, 1, 2, 3, 4, 5, 7, 10, 15, 20, 30, 50, 75, 100, 150, 200, 300, N - 1]
        print("rep %d (idle %.1f s before): " % (rep, idle) + "  ".join("%d:%.3f" % (i, ms[i]) for i in pick if i < N))
        for a, b in ((0, 5), (5, 25), (25, 100), (100, N)):
            if b <= N:
                print("   launches [%d,%d): mean %.4f ms = %.0f Msamples/s" % (a, b, sum(ms[a:b]) / (b - a), W * H * SPP / (sum(ms[a:b]) / (b - a)) / 1e3))
