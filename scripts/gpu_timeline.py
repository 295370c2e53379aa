#!/usr/bin/env python3
"""Where the time of ONE launch of the default kernel goes: per-wave time stamps (ptamd_set_timeline: kernel entry, scene
staged, no ticket left, exit) of single-frame and batched launches of the headline frame, reduced to resident waves versus
time, the spread of the waves' exit times and the share of the launch spent after the tickets ran out.
    python scripts/gpu_timeline.py [--out gpurun_out/timeline.json] [--frames 1 4] [--rows Y0:Y1]
A/B of the shared pools: PTAMD_POOL_SHARE=0 python scripts/gpu_timeline.py ..."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def reduce(tl, khz, bins_us=10.0):
    live = tl[:, 3] != 0
    t = tl[live].astype(np.float64)
    t0 = t[:, 0].min()
    us = (t - t0) / (khz / 1e3)          # ticks -> microseconds
    dry = us[:, 2][t[:, 2] != 0]         # waves that asked for a ticket and found none
    end = us[:, 3].max()
    first_dry = float(dry.min()) if len(dry) else None
    edges = np.arange(0.0, end + bins_us, bins_us)
    resident = [int(((us[:, 0] <= e) & (us[:, 3] > e)).sum()) for e in edges]
    # wave-time after the first wave found no ticket, as a share of all wave-time: what a work-conserving end would win back at most
    tail_wave_time = float(np.clip(us[:, 3] - first_dry, 0, None).sum()) if first_dry is not None else 0.0
    ideal_tail = float(len(us) * max(0.0, 0.0))
    return {
        "waves": int(live.sum()), "launch_us": round(float(end), 2),
        "staged_us_mean": round(float((us[:, 1] - us[:, 0]).mean()), 2),
        "entry_spread_us": round(float(us[:, 0].max()), 2),
        "first_dry_us": None if first_dry is None else round(first_dry, 2),
        "exit_us_percentiles_5_50_95_100": [round(float(np.percentile(us[:, 3], q)), 2) for q in (5, 50, 95, 100)],
        "tail_us": None if first_dry is None else round(float(end - first_dry), 2),
        "tail_mean_resident_fraction": None if first_dry is None else round(tail_wave_time / (len(us) * max(end - first_dry, 1e-9)), 4),
        "resident_waves_every_%gus" % bins_us: resident,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "timeline.json"))
    ap.add_argument("--frames", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--rows", default=None)
    ap.add_argument("--share", type=int, default=0, help="machine_share of the launches")
    ap.add_argument("--interleave", default=None, help="RANKS:RANK:ROWS")
    args = ap.parse_args()
    import torch
    import cuda_pathtracer_amd as P
    W, H, B = 1920, 1080, 4
    hs = P.HostScene.load(os.path.join(ROOT, "assets", "indoor.scene"))
    cube = P.cubemap_for_scene(hs)
    rows = tuple(int(v) for v in args.rows.split(":")) if args.rows else None
    ilv = tuple(int(v) for v in args.interleave.split(":")) if args.interleave else None
    out = {"tail_handover": os.environ.get("PTAMD_TAIL", "1"), "rows": rows, "interleave": ilv, "machine_share": args.share, "launches": []}
    with P.Context(0) as ctx:
        sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(cube)
        fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H, rows=rows, band_local=rows is not None, machine_share=args.share, interleave=ilv)
        n_waves = 256 * 24
        for frames in args.frames:
            for _ in range(30):   # clocks settled
                fr.render(spp=frames, bounces=B, kernel=P.KERNEL_BVH_RESTART, batched=True, reset=True)
            torch.cuda.synchronize()
            ctx.set_timeline(n_waves)
            reps = []
            for _ in range(5):
                fr.render(spp=frames, bounces=B, kernel=P.KERNEL_BVH_RESTART, batched=frames > 1, reset=True) if frames > 1 else \
                    fr.render(spp=1, bounces=B, kernel=P.KERNEL_BVH_RESTART, reset=True)
                tl, khz = ctx.read_timeline(n_waves)
                reps.append(reduce(tl, khz))
            ctx.set_timeline(0)
            reps.sort(key=lambda r: r["launch_us"])
            out["launches"].append({"frames_per_launch": frames, "clock_khz": khz, "median_of_5": reps[2]})
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    for l in out["launches"]:
        m = l["median_of_5"]
        print(l["frames_per_launch"], "frame(s):", {k: v for k, v in m.items() if not k.startswith("resident")})


if __name__ == "__main__":
    main()
