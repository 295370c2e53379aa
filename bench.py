#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s (and ms/frame) at 1920x1080, 4 spp, indoor.obj (BASELINE.json configs[1]).

One "step" = one frame of the hot path = `spp` static frames of the megakernel (frame seeds 1..spp) accumulating into
a temporal framebuffer that starts from zero (ptamd_launch.reset_accumulation), with every input resident in HBM before the timed region starts.  1 sample = one
execution of the reference kernel() for one pixel.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Started as `python bench.py --gpus N` this script launches the ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, spawned before anything touches the GPU; its
single JSON line and return code are relayed); started under torch.distributed.run it is a rank.  The frame is split
into pixel-row bands (cuda_pathtracer_amd.tiles), each rank renders its band with global-coordinate seeds, and the
finished RGBA8 bands are gathered with one RCCL all-gather per frame.  The frame is fixed, so scaling is "strong".

The headline (`value`) issues the spp frames of a step as ONE batched launch (ptamd_launch.frame_count: same
accumulator and surface as spp consecutive launches, bit for bit) and keeps two (one GPU) or four (several GPUs) steps
in flight on separate HIP streams, each launch sized to its share of the GPU, so that a step's ramp, tail, resolve pass
and all-gather overlap the bulk of the next one.  Next to it, at N = 1, the line reports
  value_unpipelined  the same batched launch issued back to back on ONE stream by a host that does not wait between frames
                     (frames_in_flight = 1; the library itself overlaps consecutive launches of a stream: DESIGN.md);
  value_sequential   SURVEY §8-d's literal definition: spp launches per frame, one after the other on one stream — the
                     reference's own call pattern (gpu_processor.cpp:365-386 issues a frame and does not wait for it);
  value_host_sync    the batched launch with a host synchronisation after every frame: nothing can overlap, `ms_per_frame`
                     is this latency; value_sequential_host_sync: the same for one launch per spp;
  other_configs      BASELINE.json configs[3] and configs[4], a few steps each.

`roofline`: the megakernel is VALU-issue / lane-divergence bound (no HBM or MFMA roof is within two orders of
magnitude: DESIGN.md §4).  bound = "valu_issue": achieved = VALU wave-instructions issued per second chip-wide during
the timed region (instructions per launch from the committed rocprofv3 PMC passes of this same workload,
profiles/pmc_latest.json, x concurrent launches / the live HIP-event duration of a launch); peak = 1024 SIMDs x 2.4 GHz
/ 2 cycles per wave64 VALU instruction.  `active_lanes` (of 64) says how many lanes an issued instruction carries; HBM
figures are secondary (`traffic`, `hbm_frac_measured`).

Rank 0 prints ONE JSON line (contract in the task statement); at N = 1 it carries `cpu_baseline` (the CPU oracle timed
on the host cores — a reported baseline, not the target).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto 4 hardware queues by default; two frames-in-flight streams sharing a queue would serialise
# (measured: -30 %; with 8 queues five frame streams collapse the same way, with 16 they do not).  Frames + RCCL's
# streams + torch's own need more than 4: ask for 16 before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

WIDTH, HEIGHT, SPP, BOUNCES = 1920, 1080, 4, 4
MAX_FRAMES_PER_LAUNCH = 4   # ptamd_api.cpp: kMaxFramesPerSlab — a batch of more frames is issued as consecutive launches of four
HBM_PEAK_GBPS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
N_SIMDS, CLOCK_GHZ, VALU_CYCLES = 1024, 2.4, 2   # 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles
VALU_PEAK_GINST = N_SIMDS * CLOCK_GHZ / VALU_CYCLES   # 1228.8 G wave-instructions/s
SCALAR_PEAK_GINST = (N_SIMDS // 4) * CLOCK_GHZ         # 614.4 G/s: one scalar instruction per clock per CU (measured 596 G/s at the clock a busy chip holds)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--bounces", type=int, default=BOUNCES)
    ap.add_argument("--scene", default=os.path.join(ROOT, "assets", "indoor.scene"))
    ap.add_argument("--atrium", action="store_true",
                    help="configs[3]: generate the seed-fixed Sponza-class atrium (264 832 triangles) as OBJ + MTL + .scene in a "
                         "temporary directory and render that scene (through the same loader as any other asset)")
    ap.add_argument("--fix-backslashes", action="store_true",
                    help="load the scene with the MTL's backslash texture paths normalised (indoor.scene then gets its real textures)")
    ap.add_argument("--tessellate", type=int, default=1, help="split every face into n*n (24 -> ~257k tris: round 1's configs[3] stand-in)")
    ap.add_argument("--aperture", type=float, default=None, help="override the camera aperture (configs[4]: 0.113)")
    ap.add_argument("--kernel", choices=["restart", "restart_fma", "persistent", "split", "bvh", "blockwise", "brute"], default="restart",
                    help="restart_fma: the opt-in contracted instantiation (NOT bit-exact: include/ptamd.h PTAMD_KERNEL_BVH_RESTART_FMA)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames rendered concurrently on separate HIP streams, each with its own buffers (double "
                         "buffering, as the reference double-buffers its GL renderbuffers: driver/interop.cpp:107-111).  "
                         "0 = auto: 2 on one GPU, 4 on several GPUs (the RCCL all-gather of frame i overlaps the render of "
                         "the next ones); each launch is sized to 1/n of the GPU (ptamd_launch.machine_share)")
    ap.add_argument("--machine-share", type=int, default=0,
                    help="size every launch to 1/K of the GPU whatever the number of frames in flight (default: K = frames in flight): "
                         "with more frames in flight than shares, the queued workgroups of one launch fill the tails and kernel gaps of the others")
    ap.add_argument("--no-share", dest="share", action="store_false",
                    help="with several frames in flight, size every launch to the whole GPU instead of its 1/n share")
    ap.add_argument("--sequential", action="store_true", help="one launch per spp instead of one batched launch per frame")
    ap.add_argument("--rows", default=None, metavar="Y0:Y1",
                    help="N = 1 only: render just surface rows [Y0, Y1) of the frame (one rank's band of a multi-GPU split, "
                         "scripts/band_proxy.py); `value` then counts the band's samples")
    ap.add_argument("--interleave", type=int, default=None, metavar="ROWS",
                    help="N > 1: ranks own interleaved bands of ROWS rows (band j -> rank j %% N) instead of one contiguous band "
                         "each; 0 = contiguous.  Default: 8 on several GPUs (contiguous eighths of the headline frame differ "
                         "by 13 %% in cost, 8-row interleaved bands by 0.7 %%: profiles/r02_band_proxy_*.json)")
    ap.add_argument("--as-rank", default=None, metavar="R/N",
                    help="one GPU only: render what rank R of an N-GPU job renders (its contiguous band, or its interleaved "
                         "bands with --interleave ROWS): the single-GPU proxy of scripts/band_proxy.py")
    ap.add_argument("--no-extra", action="store_true",
                    help="headline only: skip value_unpipelined / value_sequential / other_configs (profiling runs)")
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed rendering before the W warm-up steps of every leg, so that the timed steps run at the chip's "
                         "steady clocks (0: none)")
    ap.add_argument("--no-fallback", action="store_true", help="N > 1 started without a launcher: do not retry with the plain configuration when the ranks fail")
    ap.add_argument("--fallback-run", action="store_true", help=argparse.SUPPRESS)   # set by self_launch for its second attempt
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "pmc_latest.json"),
                    help="per-dispatch PMC averages of this workload written by scripts/summarize_pmc.py (optional)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------- N > 1 without a launcher: start the ranks ourselves

def self_launch(args) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE: run N ranks under torch.distributed.run as a CHILD process (this
    process never touches the GPU, so nothing is exec'ed or forked after HIP initialisation) and relay its output."""
    def child(extra):
        with socket.socket() as s2:
            s2.bind(("127.0.0.1", 0))
            p2 = s2.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(p2), os.path.abspath(__file__)] + sys.argv[1:] + extra
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        return r.returncode, r.stdout

    rc, line = child([])
    if rc != 0 and not args.no_fallback:
        # First contact with a multi-GPU node: the default N > 1 configuration (four frames in flight on quarter-GPU
        # launches, interleaved bands, collectives from several streams) has only ever run on one GPU.  If it fails, ONE
        # fresh set of ranks runs the plain configuration — one frame at a time, contiguous bands — and says so in its
        # line ("fallback": true).  This parent never touches the GPU; no rank is re-executed in place.
        print(f"bench.py: the {args.gpus}-rank run exited with {rc}; starting the plain configuration once", file=sys.stderr, flush=True)
        if line.strip():
            print("bench.py: (discarded output of the failed run) " + line.strip()[:2000], file=sys.stderr, flush=True)
        rc, line = child(["--frames-in-flight", "1", "--interleave", "0", "--fallback-run"])
    # stdout carries exactly one JSON line, and only that of a run that ended well
    sys.stdout.write(line if rc == 0 else "")
    if rc != 0 and line.strip():
        print("bench.py: (output of the failed run) " + line.strip()[:2000], file=sys.stderr, flush=True)
    sys.stdout.flush()
    return rc


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


KERNEL_SYMBOL = {"persistent": "pt_megakernel_persistent", "restart": "pt_megakernel_restart", "restart_fma": "ptamd_fma::pt_megakernel_restart", "blockwise": "pt_megakernel_blockwise",
                 "split": "pt_megakernel_split", "bvh": "pt_megakernel", "brute": "pt_megakernel"}


_STREAMS = {}


def frame_stream(torch, dev, i):
    """The i-th frames-in-flight stream of this process, created once and shared by every Workload: HIP deals streams to its
    hardware queues in creation order, and a bench run builds a dozen workloads (each with a context that owns two internal
    streams) — the 2 + 2 + ... streams of later workloads can land pairwise on ONE queue, where the two frames in flight then
    serialise (round 4: crate_land 14.4 -> 9.5 Gsamples/s inside the full run, 14.4 alone)."""
    key = (str(dev), i)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device=dev)
    return _STREAMS[key]


class Workload:
    """A frame configuration on this rank: scene uploaded once, `n_slots` frames in flight (buffers + stream each)."""

    def __init__(self, P, torch, dist, hs, cube, W, H, spp, B, kernel_name, n_slots, share, batched, local_rank,
                 world=1, rank=0, gather=False, rows=None, interleave=None, machine_share=0):
        """interleave = (ranks, rank, band_rows): this rank owns interleaved bands instead of rows [y0, y1)."""
        self.P, self.torch, self.dist = P, torch, dist
        self.W, self.H, self.spp, self.B = W, H, spp, B
        self.world, self.rank = world, rank
        self.kernel = {"auto": P.KERNEL_AUTO, "bvh": P.KERNEL_BVH, "brute": P.KERNEL_BRUTE_FORCE, "persistent": P.KERNEL_BVH_PERSISTENT,
                       "blockwise": P.KERNEL_BVH_BLOCKWISE, "split": P.KERNEL_BVH_SPLIT, "restart": P.KERNEL_BVH_RESTART,
                       "restart_fma": P.KERNEL_BVH_RESTART_FMA}[kernel_name]
        self.batched = batched and kernel_name in ("auto", "persistent", "split", "restart", "restart_fma") and spp > 1
        self.n_slots = n_slots
        self.hs = hs
        self.dev = torch.device("cuda", local_rank)
        self.y0, self.y1 = P.row_bands(H, world)[rank] if rows is None else rows
        self.interleave = interleave
        self.my_rows = self.y1 - self.y0 if interleave is None else P.interleaved_rows(H, *interleave)
        self.ctx = P.Context(local_rank)
        self.ctx.setup_function_tables()
        self.sid = self.ctx.upload_scene(hs)
        self.cid = self.ctx.upload_cubemap(cube)
        self.info = self.ctx.scene_info(self.sid)
        self.slots = []
        for i in range(n_slots):
            # (a proxy band is gathered as if it were the whole frame: same message size as the rank's real gather share)
            if not gather:
                bg = None
            elif world == 1:
                bg = P.BandGather(self.my_rows, W, 1, 0, self.dev)
            else:
                bg = P.BandGather(H, W, world, rank, self.dev, interleave=interleave[2] if interleave is not None else 0)
            # the renderer writes its rows straight into the collective's send buffer: no staging copy per frame
            fr = P.FrameRenderer(self.ctx, self.sid, self.cid, hs.camera_struct(), W, H, rows=(self.y0, self.y1),
                                 band_local=True, machine_share=(machine_share if machine_share > 0 else n_slots) if (share and n_slots > 1) else 0, interleave=interleave,
                                 surface=bg.send_rows() if bg is not None else None)
            st = torch.cuda.current_stream() if n_slots == 1 else frame_stream(torch, self.dev, i)
            self.slots.append((fr, bg, st))
        self.counter = 0

    @property
    def frames_per_launch(self):
        return min(self.spp, MAX_FRAMES_PER_LAUNCH) if self.batched else 1

    def step(self, ev=None, gather=True):
        torch = self.torch
        fr, bg, st = self.slots[self.counter % self.n_slots]
        self.counter += 1
        with torch.cuda.stream(st):
            if ev is not None:
                ev[0].record(st)
            # reset=True: every step starts a new accumulation (the same result as clearing the accumulator first, which
            # tests/test_gpu_parity.py asserts; the clear is folded into the launch that would read it)
            fr.render(spp=self.spp, bounces=self.B, kernel=self.kernel, stream=st, batched=self.batched, reset=True)
            if ev is not None:
                ev[1].record(st)
            if bg is not None and gather:
                bg.gather(fr.surface)  # one RCCL all-gather of the RGBA8 bands per frame

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, steps, warmup, settle_ms=0.0, sync_each=False):
        """`settle_ms` of untimed rendering (no collective: ranks need not agree on the count), W untimed steps, then
        exactly K timed steps between barriers.  Returns (wall seconds, mean ms between the HIP events that bracket one
        step's launches on its stream).  The settle phase exists because an idle MI355X needs ~30 ms of load to reach its
        steady clocks (scripts/gpu_ramp.py: launches 5..25 after 3 s of idling run 6.5 % slower than launch 30 onwards),
        longer than the W + K steps of a default run."""
        torch = self.torch
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        with _StdoutToStderr():  # first collective = communicator setup (and RCCL's banner)
            t_end = time.perf_counter() + settle_ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(self.n_slots):
                    self.step(gather=False)
                torch.cuda.synchronize()
            for _ in range(warmup):
                self.step()
            self.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(evs[i])
            if sync_each:   # a host that waits for every frame before it issues the next one
                torch.cuda.synchronize()
        self.barrier()
        dt = time.perf_counter() - t0
        self.rank_ms = None
        if self.world > 1:
            # every rank's own time (the job's is the slowest) and the all-gather on its own
            mx, mean, every = self.P.rank_times_ms(dt, self.dev)
            self.rank_ms = {"max": round(mx / steps, 4), "mean": round(mean / steps, 4), "per_rank": [round(v / steps, 4) for v in every]}
            dt = mx * 1e-3
        step_ms = sum(a.elapsed_time(b) for a, b in evs) / max(steps, 1)
        return dt, step_ms

    def gather_ms(self, reps=20):
        """One all-gather of the finished bands with nothing else in flight, mean of `reps` (None without a collective)."""
        bg = self.slots[0][1]
        if bg is None:
            return None
        with _StdoutToStderr():
            return self.P.time_gather_ms(bg, reps, self.torch.cuda.synchronize)

    def trace_stats(self):
        """Exact traversal counts of one frame (instrumented build of the same kernel, untimed)."""
        P = self.P
        stats = {k: 0 for k in ("rays", "nodes_visited", "tris_tested", "mesh_hits", "nmap_hits", "samples",
                                "wave_node_iters", "idle_unstarted", "idle_finished", "idle_parked")}
        scratch = P.FrameRenderer(self.ctx, self.sid, self.cid, self.hs.camera_struct(), self.W, self.H,
                                  rows=(self.y0, self.y1), band_local=True, interleave=self.interleave)
        # the same launches as the timed ones: one batched launch of spp frames, or spp single-frame launches
        per_launch = self.frames_per_launch
        for k in range(1, self.spp + 1, per_launch):
            l = self.ctx.make_launch(scratch.surface, scratch.accum, self.sid, self.cid, self.hs.camera_struct(), self.W,
                                     self.H, frame_nb=k, bounces=self.B, rows=scratch.rows,
                                     kernel=P.KERNEL_BVH_RESTART if self.kernel == P.KERNEL_BVH_RESTART_FMA else self.kernel,   # (the contracted kernel has no instrumented build)
                                     band_local_buffers=True, interleave=self.interleave, frame_count=per_launch)
            s = self.ctx.raytrace_stats(l)
            for key in stats:
                stats[key] += s[key]
        self.torch.cuda.synchronize()
        return stats

    def checksum(self):
        return int(self.slots[0][0].surface.to(self.torch.int64).sum().item())

    def close(self):
        self.slots = []
        self.ctx.close()


def load_pmc(path, kernel_name, W, H, spp, B, frames_per_launch, tessellate, scene, build_id):
    """A PMC record (scripts/summarize_pmc.py) applies only to the workload it was collected on, and only to the build whose id it
    carries: counters of another build are marked stale and not used (frac / achieved / traffic stay null)."""
    try:
        with open(path) as f:
            pj = json.load(f)
    except (OSError, ValueError):
        return None
    pj["stale"] = pj.get("build_id") != build_id
    ok = (pj.get("kernel") == kernel_name and pj.get("workload") == f"{W}x{H}" and pj.get("spp") == spp
          and pj.get("bounces") == B and pj.get("frames_per_launch") == frames_per_launch and tessellate == 1
          and pj.get("scene", "indoor.scene") == scene)
    return pj if ok else None


def roofline_block(pmc, kernel_symbol, kern_ms, n_slots, samples_per_launch, compulsory_launch, build_id, with_traffic=True):
    """The VALU-issue roofline of one workload: achieved = VALU wave-instructions per launch (PMC record of this workload and this
    build) x concurrent launches / the live HIP-event duration of a launch; beside it what the counters say about the memory side
    (bytes read beyond L2 against the compulsory bytes, requests beyond L1, share of wave cycles spent waiting)."""
    stale = bool(pmc is not None and pmc.get("stale"))
    if stale:
        pmc = None
    g = (lambda k: None) if pmc is None else pmc.get
    valu_per_sample, active_lanes = g("valu_insts_per_sample"), g("active_lanes")
    traffic = g("hbm_bytes_per_launch") if with_traffic else None
    valu_launch = None if valu_per_sample is None else valu_per_sample * samples_per_launch
    achieved = None if valu_launch is None else valu_launch * n_slots / (kern_ms * 1e-3) / 1e9
    frac = None if achieved is None else achieved / VALU_PEAK_GINST
    per_s = n_slots / (kern_ms * 1e-3) / 1e9
    return {
        "bound": "valu_issue", "achieved": None if achieved is None else round(achieved, 2),
        "peak": VALU_PEAK_GINST, "unit": "G wave-inst/s", "frac": None if frac is None else round(frac, 4),
        "traffic": traffic,
        "kernel": kernel_symbol, "kernel_ms_per_launch": round(kern_ms, 4), "concurrent_launches": n_slots,
        "samples_per_launch": int(samples_per_launch),
        "valu_insts_per_launch": None if valu_launch is None else int(valu_launch),
        "valu_insts_per_sample": valu_per_sample,
        "active_lanes": active_lanes,   # SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU: lanes (of 64) per issued VALU instruction
        "useful_lane_frac": None if (frac is None or active_lanes is None) else round(frac * active_lanes / 64.0, 4),
        # the same instructions against the clocks the profiled dispatch actually ran for: share of the VALU issue slots of the
        # launch's 1/concurrent_launches of the GPU
        "valu_busy_at_measured_clock": None if (pmc is None or not g("gui_active_cycles_per_launch") or not g("valu_insts_per_launch"))
        else round(g("valu_insts_per_launch") * VALU_CYCLES * n_slots / (N_SIMDS * g("gui_active_cycles_per_launch")), 4),
        # the CU's ONE scalar unit (SALU + branches; one instruction per clock per CU = half the VALU rate of its four SIMDs, measured:
        # scripts/ubench/ifetch_rate.hip, profiles/r04_issue_ubench_*.txt) — the second issue roof of divergent compiled code
        "scalar_issue": None if (pmc is None or g("salu_insts_per_launch") is None) else {
            "salu_insts_per_launch": int(g("salu_insts_per_launch")),
            "branch_insts_per_launch": None if g("branch_insts_per_launch") is None else int(g("branch_insts_per_launch")),
            "achieved": round((g("salu_insts_per_launch") + (g("branch_insts_per_launch") or 0.0)) * per_s, 2),
            "peak": SCALAR_PEAK_GINST, "unit": "G wave-inst/s",
            "frac": round((g("salu_insts_per_launch") + (g("branch_insts_per_launch") or 0.0)) * per_s / SCALAR_PEAK_GINST, 4)},
        "pmc_source": g("source"), "build_id": build_id, "pmc_stale": stale,
        "peak_note": f"{N_SIMDS} SIMDs x {CLOCK_GHZ} GHz / {VALU_CYCLES} cycles per wave64 VALU instruction "
                     "(MI355X_MICROARCH.md: v_fma_f32 2 cyc on SIMD-32; 64 lanes x 2 flop x this = the 157.3 TFLOP/s FP32 vector peak)",
        "hbm_peak_gbps": HBM_PEAK_GBPS,
        "compulsory_hbm_bytes_per_launch": int(compulsory_launch),
        "compulsory_hbm_gbps": round(compulsory_launch * per_s, 2),
        "hbm_gbps_measured": None if traffic is None else round(traffic * per_s, 2),
        "hbm_frac_measured": None if traffic is None else round(traffic * per_s / HBM_PEAK_GBPS, 4),
        "refetch_factor": None if (traffic is None or not compulsory_launch) else round(traffic / compulsory_launch, 2),
        # walks served from the caches: requests beyond L1 per sample, their mean latency, L2 hit rate, share of wave cycles waiting
        "l1_to_l2_requests_per_sample": None if g("l1_to_l2_requests_per_launch") is None else round(g("l1_to_l2_requests_per_launch") / samples_per_launch, 2),
        "l1_to_l2_mean_latency_cycles": None if g("l1_to_l2_mean_latency_cycles") is None else round(g("l1_to_l2_mean_latency_cycles"), 1),
        "l2_hit_rate": None if (g("l2_hits_per_launch") is None or g("l2_misses_per_launch") is None)
        else round(g("l2_hits_per_launch") / max(g("l2_hits_per_launch") + g("l2_misses_per_launch"), 1.0), 4),
        "wait_any_share_of_wave_cycles": None if g("wait_any_share_of_wave_cycles") is None else round(g("wait_any_share_of_wave_cycles"), 4),
    }


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    import cuda_pathtracer_amd as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    tmp_dirs = []

    def atrium_scene():
        import tempfile
        from cuda_pathtracer_amd.synthetic import write_atrium
        tmp_dirs.append(tempfile.TemporaryDirectory(prefix="ptamd_atrium_"))
        return write_atrium(tmp_dirs[-1].name)

    if args.atrium:
        args.scene = atrium_scene()
    hs = P.HostScene.load(args.scene, normalise_backslashes=args.fix_backslashes)
    if args.tessellate > 1:
        hs = P.tessellate(hs, args.tessellate)
    if args.aperture is not None:
        hs.camera["aperture"] = args.aperture
    # (indoor.scene names a cube cross that does not ship: the 1x1 fallback colour either way, as in the reference)
    cube = P.cubemap_for_scene(hs, asset_folder=os.path.dirname(os.path.abspath(args.scene)))
    # measured on one rank's share of an 8-way split (scripts/band_proxy.py): 3 / 4 / 5 frames in flight = 0.166 / 0.155 / 0.183 ms
    # whole frame on one GPU, last build of round 2: 2 / 3 / 4 in flight = 9927 / 9930 / 10031 Msamples/s over 60 steps
    # (scripts/gpu_fif.sh) but 9815 / 9660 / 9590 over the 20 steps the driver times: the deeper pipeline drains longer at the
    # end of the timed region than it gains in steady state
    n_slots = args.frames_in_flight if args.frames_in_flight > 0 else (4 if (world > 1 or args.as_rank is not None) else 2)
    gather = world > 1 or force_gather

    rows = None
    interleave = None
    if args.rows is not None:
        if world != 1:
            raise SystemExit("--rows applies to one GPU only")
        rows = tuple(int(v) for v in args.rows.split(":"))
        if not (0 <= rows[0] < rows[1] <= H):
            raise SystemExit("--rows Y0:Y1 must lie inside the frame")
    ilv_rows = args.interleave if args.interleave is not None else (8 if (world > 1 and args.kernel == "restart") else 0)
    if args.as_rank is not None:
        if world != 1:
            raise SystemExit("--as-rank applies to one GPU only")
        pr, pn = (int(v) for v in args.as_rank.split("/"))
        if ilv_rows:
            interleave = (pn, pr, ilv_rows)
        else:
            rows = P.row_bands(H, pn)[pr]
    elif world > 1 and ilv_rows:
        interleave = (world, rank, ilv_rows)
    wl = Workload(P, torch, dist, hs, cube, W, H, spp, B, args.kernel, n_slots, args.share, not args.sequential,
                  local_rank, world, rank, gather, rows, interleave, machine_share=args.machine_share)
    proxy = rows is not None or (interleave is not None and world == 1)
    dt, step_ms = wl.run(args.steps, args.warmup, args.settle_ms)
    gather_ms = wl.gather_ms() if (world > 1 or force_gather) else None
    frames_per_launch = wl.frames_per_launch
    launches_per_step = -(-spp // frames_per_launch)
    # one step = launches_per_step megakernel launches (+ the small resolve kernel when batched) back to back on its stream
    kern_ms = step_ms / launches_per_step
    value = W * (wl.my_rows if proxy else H) * spp * args.steps / dt / 1e6
    info = wl.info
    stats = wl.trace_stats()
    checksum = wl.checksum()
    y0, y1 = wl.y0, wl.y1
    samples_per_launch = wl.my_rows * W * frames_per_launch
    compulsory_launch = 28 * wl.my_rows * W * frames_per_launch + hs.scene_bytes()

    extra = {}
    is_headline = (W, H, spp, B, args.tessellate, args.aperture, args.atrium, proxy, args.fix_backslashes, os.path.basename(args.scene)) == \
                  (WIDTH, HEIGHT, SPP, BOUNCES, 1, None, False, False, False, "indoor.scene")
    if world == 1 and not args.no_extra and not force_gather and not proxy:
        k2 = max(4, args.steps)   # (as many steps as the headline: a whole-GPU launch opens every back-to-back sequence, short regions overweigh it)
        # (a) the same batched launch on one stream: issued back to back, and with the host waiting for every frame (latency)
        solo = Workload(P, torch, dist, hs, cube, W, H, spp, B, args.kernel, 1, False, not args.sequential, local_rank)
        sdt, s_ms = solo.run(k2, 2, args.settle_ms)
        extra["value_unpipelined"] = round(W * H * spp * k2 / sdt / 1e6, 3)
        hdt, h_ms = solo.run(k2, 2, 0.0, sync_each=True)
        solo.close()
        extra["value_host_sync"] = round(W * H * spp * k2 / hdt / 1e6, 3)
        extra["ms_per_frame"] = round(h_ms, 4)
        # (b) SURVEY §8-d's literal metric: spp launches per frame on one stream
        if not args.sequential:
            # (PTAMD_KERNEL_AUTO: what a host that calls raytrace() once per spp gets)
            seq = Workload(P, torch, dist, hs, cube, W, H, spp, B, "auto" if args.kernel == "restart" else args.kernel, 1, False, False, local_rank)
            qdt, q_ms = seq.run(k2, 2, args.settle_ms)
            extra["value_sequential"] = round(W * H * spp * k2 / qdt / 1e6, 3)
            extra["ms_per_frame_sequential"] = round(q_ms, 4)
            qdt, q_ms = seq.run(k2, 2, 0.0, sync_each=True)
            seq.close()
            extra["value_sequential_host_sync"] = round(W * H * spp * k2 / qdt / 1e6, 3)
        # (b') the opt-in contracted instantiation on the same workload: NOT bit-exact (held to BASELINE.md section 5's tolerance by the
        # GPU suite), reported beside `value`, never as it
        if args.kernel == "restart":
            fw = Workload(P, torch, dist, hs, cube, W, H, spp, B, "restart_fma", n_slots, args.share, not args.sequential, local_rank)
            fdt, f_ms = fw.run(args.steps, args.warmup, args.settle_ms)
            fw.close()
            extra["value_fma"] = round(W * H * spp * args.steps / fdt / 1e6, 3)
            extra["value_fma_note"] = ("PTAMD_KERNEL_BVH_RESTART_FMA: the same kernel with floating-point contraction allowed (what the reference's nvcc "
                                       "build permits); not bit-exact — all but <= 1e-4 of the pixels identical, mean image within 2e-6 (BASELINE.md section 5)")
        # (c) the other single-GPU configurations of BASELINE.json, a few steps each
        if is_headline and args.kernel == "restart":
            others = []
            def load_textured_indoor():
                return P.HostScene.load(args.scene, normalise_backslashes=True)

            def load_crate_land():
                return P.HostScene.load(os.path.join(ROOT, "assets", "crate_land.scene"))

            for name, loader, (w2, h2, s2, b2), ap, k, pmc_file, pmc_scene in (
                    ("configs[3]: atrium.obj (generated Sponza-class OBJ, 264 832 triangles, through the loader; walked from L2) "
                     "1920x1080 4 spp 4 bounces", lambda: P.HostScene.load(atrium_scene()), (1920, 1080, 4, 4), None, 6, "pmc_atrium.json", "atrium.scene"),
                    ("configs[4]: indoor.scene 3840x2160 16 spp 8 bounces aperture 0.113",
                     lambda: P.HostScene.load(args.scene), (3840, 2160, 16, 8), 0.113, 3, "pmc_c4k.json", "indoor.scene aperture 0.113"),
                    # the texture path (sampleTexture + normal maps, intersection.cuh:20-65,216-242): dependent 16 B / 12 B gathers
                    ("textured indoor.scene: the maps indoor.mtl names (backslash paths normalised; 1024^2 parquet / concrete albedo + normal "
                     "maps, wooden_planck, crack2: 66 MB of float texels) 1920x1080 4 spp 4 bounces", load_textured_indoor, (1920, 1080, 4, 4), None, 8, "pmc_textured_indoor.json", "indoor.scene (textured)"),
                    ("crate_land.scene: 1024^2 RGBA + normal maps, bilinear 1024^2 cubemap (field_with_house.jpg), aperture 0.113, "
                     "1920x1080 4 spp 4 bounces", load_crate_land, (1920, 1080, 4, 4), None, 8, "pmc_crate_land.json", "crate_land.scene")):
                sc = loader()
                if ap is not None:
                    sc.camera["aperture"] = ap
                o = Workload(P, torch, dist, sc, P.cubemap_for_scene(sc, asset_folder=os.path.join(ROOT, "assets")), w2, h2, s2, b2, "restart", 2, True, True, local_rank)
                odt, o_ms = o.run(k, 1, args.settle_ms)
                oi = o.info
                o.close()
                # the counters of THIS workload on THIS build (scripts/gpu_round4.sh collects one record per bench configuration);
                # a record of another build prices nothing: pmc_stale
                bid = P.native.load().ptamd_build_id().decode()
                fpl2 = min(s2, MAX_FRAMES_PER_LAUNCH)          # frames per megakernel launch, launches per step
                lps2 = -(-s2 // fpl2)
                opmc = load_pmc(os.path.join(os.path.dirname(args.pmc_json), pmc_file), "restart", w2, h2, s2, b2, fpl2, 1, pmc_scene, bid)
                others.append({"workload": name, "value": round(w2 * h2 * s2 * k / odt / 1e6, 3), "unit": "Msamples/s",
                               "steps": k, "ms_per_step": round(odt / k * 1e3, 4), "kernel_ms_per_launch": round(o_ms / lps2, 4),
                               "launches_per_step": lps2, "frames_per_launch": fpl2,
                               "frames_in_flight": 2, "faces": oi["n_faces"], "bvh_nodes": oi["n_nodes"],
                               "roofline": roofline_block(opmc, KERNEL_SYMBOL["restart"], o_ms / lps2, 2, w2 * h2 * fpl2, 28 * w2 * h2 * fpl2 + sc.scene_bytes(), bid)})
            extra["other_configs"] = others

    if rank == 0:
        build_id = P.native.load().ptamd_build_id().decode()
        scene_tag = "atrium.scene" if args.atrium else os.path.basename(args.scene) + (" (textured)" if args.fix_backslashes else "")
        pmc = load_pmc(args.pmc_json, args.kernel, W, H, spp, B, frames_per_launch, args.tessellate, scene_tag, build_id)
        box_iters = stats["wave_node_iters"]
        roof = roofline_block(pmc, KERNEL_SYMBOL[args.kernel], kern_ms, n_slots, samples_per_launch, compulsory_launch, build_id, with_traffic=world == 1)
        roof.update({
            "nodes_per_ray": round(stats["nodes_visited"] / max(stats["rays"], 1), 2),
            "tris_per_ray": round(stats["tris_tested"] / max(stats["rays"], 1), 2),
            "rays_per_sample": round(stats["rays"] / max(stats["samples"], 1), 3),
            # instrumented build of the same kernel: share of the box-test loop's lane slots that test a box
            "box_loop_lane_utilisation": None if box_iters == 0 else round(stats["nodes_visited"] / (64.0 * box_iters), 4),
        })
        # SURVEY §8(d)'s per-unit figure, re-homed: the traversal's algorithmic bytes are LDS reads on this scene (36 B per node
        # visit: 32-byte box + 4-byte link word; 48 B per triangle record), counted exactly by the instrumented build, plus the
        # 28 B of accumulator / surface traffic per sample; against the aggregate LDS read rate (MI355X_MICROARCH.md: ~150 TB/s for
        # ds_read_b64/b128 with every CU streaming).  It documents that LDS bandwidth does not bind either.
        if info["lds_bytes_bvh"] <= 64 * 1024 and stats["samples"]:
            trav = (stats["nodes_visited"] * 36.0 + stats["tris_tested"] * 48.0) / stats["samples"]
            lds_gbps = trav * W * (wl.my_rows if proxy else H) * spp * args.steps / dt / 1e9
            roof["lds"] = {"bound": "lds", "algorithmic_bytes_per_sample": round(trav + 28.0, 1), "traversal_lds_bytes_per_sample": round(trav, 1),
                           "achieved": round(lds_gbps, 1), "peak": 150000.0, "unit": "GB/s", "frac": round(lds_gbps / 150000.0, 4)}
        cfg_tag = ""
        if is_headline:
            cfg_tag = " (configs[1])" if world == 1 else " (configs[2])"
        out = {
            "metric": f"Msamples/sec at {W}x{H}, {spp} spp, {os.path.basename(args.scene).replace('.scene', '.obj')}",
            "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            # latency of one frame (launch -> resolved surface on its stream); with frames in flight a step completes
            # every ms_per_step but each frame takes longer than that
            "ms_per_frame": extra.pop("ms_per_frame", round(step_ms, 4)),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic: assets/indoor.scene (reference asset; 1x1 textures, constant env as the reference loads it on Linux)",
            "config": {"workload": f"{os.path.basename(args.scene)}" + (f" x{args.tessellate}^2 tessellation" if args.tessellate > 1 else "")
                                   + (" (generated, 264 832 triangles)" if args.atrium else "")
                                   + f" {W}x{H} {spp} spp {B} bounces" + cfg_tag
                                   + (f", rows [{rows[0]},{rows[1]}) only" if rows is not None else "")
                                   + (f", interleaved {interleave[2]}-row bands of rank {interleave[1]}/{interleave[0]} only" if (proxy and interleave) else ""),
                       "kernel": args.kernel, "launches_per_frame": launches_per_step, "frames_per_launch": frames_per_launch,
                       "faces": info["n_faces"], "bvh_nodes": info["n_nodes"], "frames_in_flight": n_slots,
                       # untimed rendering ahead of the W warm-up steps (the chip's clocks settle ~30 ms after an idle
                       # period: scripts/gpu_ramp.py); the timed region is exactly `steps` steps
                       "untimed_settle_ms": args.settle_ms,
                       "parallelism": (f"rows/{world}" if interleave is None or world == 1 else f"interleaved {interleave[2]}-row bands over {world} ranks")
                                      + (" + RCCL all-gather of RGBA8 bands" if world > 1 else "")},
            "roofline": roof,
            "rgba_checksum_rank0_band": checksum,
        }
        if gather_ms is not None:
            # the gather is inside ms_per_step (overlapped with the next frames' rendering) and reported on its own
            out["gather_ms"] = round(gather_ms, 4)
        if wl.rank_ms is not None:
            out["per_rank_ms"] = wl.rank_ms          # ms per step on each rank's own clock: max = the job's, max / mean = imbalance
        if args.fallback_run:
            out["fallback"] = True
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hs, cube, W, H, B, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    bg = wl.slots[0][1]
    if bg is not None and rank == 0:
        # the gathered frame must equal rank 0's own band in its rows (cheap self-check of the collective)
        frame = bg.assemble()
        mine = wl.slots[0][0].surface
        if world == 1:                             # (a proxy band is gathered as a frame of its own)
            ok = torch.equal(frame[:wl.my_rows], mine)
        elif interleave is not None:
            ok, local = True, 0
            for b, e in P.interleaved_bands(H, world, rank, interleave[2]):
                ok = ok and torch.equal(frame[b:e], mine[local:local + (e - b)])
                local += e - b
        else:
            ok = torch.equal(frame[y0:y1], mine)
        assert ok, "gathered frame does not contain rank 0's rows"
    if world > 1 or force_gather:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
