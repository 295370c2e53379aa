#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s (and ms/frame) at 1920x1080, 4 spp, indoor.obj.

One "step" = one frame of the hot path = `spp` consecutive static launches of the megakernel
(frame seeds 1..spp) accumulating into a zeroed temporal framebuffer, with every input
resident in HBM before the timed region starts.  1 sample = one execution of the reference
kernel() for one pixel.

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU): the frame is split into
contiguous pixel-row bands (cuda_pathtracer_amd.tiles), each rank renders its band with
global-coordinate seeds, and the finished RGBA8 bands are gathered with one RCCL all-gather
per frame.  The frame is fixed, so scaling is "strong".

Frames are pipelined: two (one GPU) or three (several GPUs) frames are in flight on separate HIP streams, each
launch sized to its share of the GPU (ptamd_launch.machine_share), so that a frame's ramp, tail, resolve pass
and all-gather overlap the bulk of the next one — a launch costs 0.11 ms + 0.28 ms per frame, and
that fixed part is what the overlap hides.  All K timed steps start and finish inside the timed region.  A launch
that shares the GPU lasts about twice its share of the step: `roofline.kernel_ms_per_launch` is that measured
duration (it is what rocprofv3 reports too), `roofline.concurrent_launches` says how many run side by side.
--frames-in-flight 1 gives the unpipelined figure.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and, at N = 1,
`cpu_baseline` (the CPU oracle timed on the host cores — a reported baseline, not the target).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto 4 hardware queues by default; two frames-in-flight streams sharing a queue would serialise
# (measured: -30 %).  Frames + the RCCL stream + torch's own streams need more than 4: ask for 8 before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

WIDTH, HEIGHT, SPP, BOUNCES = 1920, 1080, 4, 4
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--bounces", type=int, default=BOUNCES)
    ap.add_argument("--scene", default=os.path.join(ROOT, "assets", "indoor.scene"))
    ap.add_argument("--tessellate", type=int, default=1, help="split every face into n*n (24 -> ~257k tris, configs[3])")
    ap.add_argument("--aperture", type=float, default=None, help="override the camera aperture (configs[4]: 0.113)")
    ap.add_argument("--kernel", choices=["persistent", "split", "bvh", "blockwise", "brute"], default="persistent")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames rendered concurrently on separate HIP streams, each with its own context and "
                         "buffers (double buffering, as the reference double-buffers its GL renderbuffers: "
                         "driver/interop.cpp:107-111).  0 = auto: 2 on one GPU, 3 on several GPUs (the RCCL "
                         "all-gather of frame i overlaps the render of frames i+1, i+2); each launch is sized to 1/n of "
                         "the GPU (ptamd_launch.machine_share) so that the launches co-reside")
    ap.add_argument("--no-share", dest="share", action="store_false",
                    help="with several frames in flight, size every launch to the whole GPU instead of its 1/n share")
    ap.add_argument("--sequential", action="store_true", help="one launch per spp instead of one batched launch per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="PMC-derived HBM bytes per launch written by scripts/collect_traffic.py (optional)")
    return ap.parse_args()


class _StdoutToStderr:
    """Routes fd 1 to fd 2 while active: RCCL prints a version banner to stdout on first use, and
    stdout must carry exactly one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask and cgroup CPU quota (the GPU box exposes
    every host core but grants a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(hs, cube, width, height, bounces, target_seconds):
    """The oracle (a port of the reference algorithm: brute force over every face) on all host
    cores, on a bounded sample of the SAME workload: whole-frame 1-spp launches (seeds 1, 2, ..)
    until ~target_seconds of wall time or `SPP` launches."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import pt_oracle as O
    cores = usable_cores()
    osc = O.OracleScene.from_host_scene(hs, cube)
    ocam = O.camera_from_record(hs.camera)
    # rows sample sized from a quick probe so that slow hosts stay bounded
    t0 = time.perf_counter()
    O.render(osc, ocam, width, height, spp=1, bounces=bounces, rows=(height // 2, height // 2 + 8), nthreads=cores)
    probe = max(time.perf_counter() - t0, 1e-4)
    rows_per_sec = 8.0 / probe
    rows = int(min(height, max(16, rows_per_sec * target_seconds / 1.0)))
    launches = 1
    if rows >= height:
        rows = height
        launches = int(max(1, min(SPP, target_seconds // max(height / rows_per_sec, 1e-3))))
    y0 = (height - rows) // 2
    acc = np.zeros((height, width, 3), dtype=np.float32)
    t0 = time.perf_counter()
    O.render(osc, ocam, width, height, spp=launches, bounces=bounces, rows=(y0, y0 + rows), nthreads=cores, accum=acc)
    dt = time.perf_counter() - t0
    samples = rows * width * launches
    return {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"oracle/pt_oracle.c (brute-force reference algorithm), rows [{y0},{y0 + rows}) of the "
                      f"{width}x{height} frame, {launches} launch(es) of 1 spp, {bounces} bounces, "
                      f"{cores} threads, {dt:.1f} s"}


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import cuda_pathtracer_amd as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    force_gather = os.environ.get("PTAMD_BENCH_FORCE_GATHER") == "1"  # exercise the collective at N = 1
    if world > 1 or force_gather:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        with _StdoutToStderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W, H, spp, B = args.width, args.height, args.spp, args.bounces
    kernel = {"bvh": P.KERNEL_BVH, "brute": P.KERNEL_BRUTE_FORCE, "persistent": P.KERNEL_BVH_PERSISTENT,
              "blockwise": P.KERNEL_BVH_BLOCKWISE, "split": P.KERNEL_BVH_SPLIT}[args.kernel]
    hs = P.HostScene.load(args.scene)
    if args.tessellate > 1:
        hs = P.tessellate(hs, args.tessellate)
    if args.aperture is not None:
        hs.camera["aperture"] = args.aperture
    cube = P.cubemap_for_scene(hs)
    n_slots = args.frames_in_flight if args.frames_in_flight > 0 else (3 if world > 1 else 2)
    dev = torch.device("cuda", local_rank)
    y0, y1 = P.row_bands(H, world)[rank]
    batched = (not args.sequential) and args.kernel in ("persistent", "split") and spp > 1

    class Slot:
        """One frame in flight: its own context (ticket counters, sample scratch), buffers and stream."""

        def __init__(self, use_current_stream):
            self.ctx = P.Context(local_rank)
            self.ctx.setup_function_tables()
            self.sid = self.ctx.upload_scene(hs)
            self.cid = self.ctx.upload_cubemap(cube)
            self.fr = P.FrameRenderer(self.ctx, self.sid, self.cid, hs.camera_struct(), W, H, rows=(y0, y1), band_local=True,
                                      machine_share=n_slots if args.share else 0)
            self.bg = P.BandGather(H, W, world, rank, dev) if (world > 1 or force_gather) else None
            self.stream = torch.cuda.current_stream() if use_current_stream else torch.cuda.Stream(device=dev)

    slots = [Slot(n_slots == 1) for _ in range(n_slots)]
    ctx, sid, cid, fr, bg = slots[0].ctx, slots[0].sid, slots[0].cid, slots[0].fr, slots[0].bg
    info = ctx.scene_info(sid)

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    counter = [0]

    def step(i_timed=None):
        sl = slots[counter[0] % n_slots]
        counter[0] += 1
        with torch.cuda.stream(sl.stream):
            sl.fr.accum.zero_()
            if i_timed is not None:
                ev0[i_timed].record(sl.stream)
            sl.fr.render(spp=spp, bounces=B, kernel=kernel, stream=sl.stream, batched=batched)
            if i_timed is not None:
                ev1[i_timed].record(sl.stream)
            if sl.bg is not None:
                sl.bg.gather(sl.fr.surface)  # one RCCL all-gather of the RGBA8 bands per frame

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with _StdoutToStderr():  # first collective = communicator setup (and RCCL's banner)
        for _ in range(args.warmup):
            step()
        barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # kernel time from HIP events on the launch stream (average megakernel launch duration)
    # one launch = `frames_per_launch` 1-spp frames (batched: all spp in one megakernel dispatch + the small
    # resolve kernel; sequential: one frame)
    frames_per_launch = spp if batched else 1
    kern_ms = sum(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / max(args.steps, 1) / (spp / frames_per_launch)

    # exact traversal counts for the algorithmic-bytes figure (instrumented build, untimed)
    stats = {k: 0 for k in ("rays", "nodes_visited", "tris_tested", "mesh_hits", "nmap_hits", "samples")}
    scratch = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), W, H, rows=(y0, y1), band_local=True)
    for k in range(1, spp + 1):
        l = ctx.make_launch(scratch.surface, scratch.accum, sid, cid, hs.camera_struct(), W, H, frame_nb=k,
                            bounces=B, rows=(y0, y1), kernel=kernel, band_local_buffers=True)
        s = ctx.raytrace_stats(l)
        for key in stats:
            stats[key] += s[key]
    torch.cuda.synchronize()
    # bytes/sample = 28 + 16*h_mesh + 12*h_nmap + T,
    #   T(BVH) = nodes_visited*64 + tris_tested*48 + rays*n_lights*32 ; T(brute) = rays*(n_faces*36 + n_lights*32)
    n_lights = info["n_lights"]
    if kernel != P.KERNEL_BRUTE_FORCE:
        trav = stats["nodes_visited"] * info["node_bytes"] + stats["tris_tested"] * info["tri_bytes"] + stats["rays"] * n_lights * 32
    else:
        trav = stats["rays"] * (info["n_faces"] * 36 + n_lights * 32)
    alg_bytes_frame = 28 * stats["samples"] + 16 * stats["mesh_hits"] + 12 * stats["nmap_hits"] + trav
    alg_bytes_launch = alg_bytes_frame / spp * frames_per_launch
    achieved = alg_bytes_launch / (kern_ms * 1e-3) / 1e9
    compulsory_launch = (28 * (y1 - y0) * W) * frames_per_launch + hs.scene_bytes()

    checksum = int(fr.surface.to(torch.int64).sum().item())
    samples_total = W * H * spp * args.steps
    value = samples_total / dt / 1e6

    if rank == 0:
        traffic = None
        if os.path.exists(args.traffic_json):
            try:
                with open(args.traffic_json) as f:
                    tj = json.load(f)
                if (tj.get("kernel") == args.kernel and tj.get("workload") == f"{W}x{H}" and args.tessellate == 1
                        and tj.get("frames_per_launch") == frames_per_launch and tj.get("bounces") == B):
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/sec at 1920x1080, 4 spp, indoor.obj",
            "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_frame": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic: assets/indoor.scene (reference asset; 1x1 textures, constant env as the reference loads it on Linux)",
            "config": {"workload": f"{os.path.basename(args.scene)}" + (f" x{args.tessellate}^2 tessellation" if args.tessellate > 1 else "")
                                   + f" {W}x{H} {spp} spp {B} bounces"
                                   + (f" ({'configs[1]' if world == 1 else 'configs[2]'})" if (W, H, spp, B, args.tessellate, args.aperture) == (WIDTH, HEIGHT, SPP, BOUNCES, 1, None) else ""),
                       "kernel": args.kernel, "launches_per_frame": 1 if batched else spp, "frames_per_launch": frames_per_launch, "faces": info["n_faces"], "bvh_nodes": info["n_nodes"],
                       "frames_in_flight": n_slots,
                       "parallelism": f"rows/{world}" + (" + RCCL all-gather of RGBA8 bands" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         # launches in flight co-reside (each on 1/n of the GPU): a launch lasts ~n times its share of the
                         # step; `achieved` is per launch as defined, the chip-wide rate is n times that
                         "concurrent_launches": n_slots, "achieved_all_launches": round(achieved * n_slots, 2),
                         "kernel": {"persistent": "pt_megakernel_persistent", "blockwise": "pt_megakernel_blockwise", "split": "pt_megakernel_split"}.get(args.kernel, "pt_megakernel"), "kernel_ms_per_launch": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": int(alg_bytes_launch),
                         "samples_per_launch": int(stats["samples"] / spp * frames_per_launch),
                         "algorithmic_bytes_per_sample": round(alg_bytes_frame / max(stats["samples"], 1), 1),
                         "compulsory_hbm_bytes_per_launch": int(compulsory_launch),
                         "compulsory_hbm_gbps": round(compulsory_launch / (kern_ms * 1e-3) / 1e9, 2),
                         # HBM bytes the PMC counters saw for this kernel (`traffic`) over its measured duration
                         "hbm_gbps_measured": None if traffic is None else round(traffic / (kern_ms * 1e-3) / 1e9, 2),
                         "hbm_frac_measured": None if traffic is None else round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "note": "traversal bytes are served from LDS, not HBM (DESIGN.md 'Roofline'); "
                                 + (f"{n_slots} launches run side by side, each on 1/{n_slots} of the GPU: `achieved` and `kernel_ms_per_launch` "
                                    "are per launch, `achieved_all_launches` is the chip-wide rate" if n_slots > 1 else "one launch at a time"),
                         "nodes_per_ray": round(stats["nodes_visited"] / max(stats["rays"], 1), 2),
                         "tris_per_ray": round(stats["tris_tested"] / max(stats["rays"], 1), 2),
                         "rays_per_sample": round(stats["rays"] / max(stats["samples"], 1), 3)},
            "rgba_checksum_rank0_band": checksum,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hs, cube, W, H, B, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if bg is not None and rank == 0:
        # the gathered frame must equal rank 0's own band in its rows (cheap self-check of the collective)
        frame = bg.assemble()
        assert torch.equal(frame[y0:y1], fr.surface), "gathered frame does not contain rank 0's band"
    if world > 1 or force_gather:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
