#!/usr/bin/env python3
"""configs[3], round 4: how does the four-wide walk scale with waves per SIMD when it carries no path state?

The restart megakernel walks the atrium at 4 waves per SIMD (128 VGPRs: path state + walk) and every attempt to relieve ONE of its
shared resources moved it by ~1 % (profiles/r04_notes.md).  This script measures the walk alone — ptamd_trace_rays_queue: persistent
waves pulling rays from a queue, the restart kernel's own visit code, lanes refilled as soon as a few are idle — on a ray set that
mimics the megakernel's: the 1080p primary rays of the atrium's camera and three generations of diffuse bounce rays from their hit
points (cosine-distributed about the geometric normal of the face hit), in path order.

    python3 scripts/gpu_trace_queue.py   ->  JSON on stdout (rays per second per configuration, records identical across configurations)
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import cuda_pathtracer_amd as P  # noqa: E402
from cuda_pathtracer_amd.synthetic import write_atrium  # noqa: E402

W, H = 1920, 1080
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory(prefix="ptamd_atrium_") as d:
    hs = P.HostScene.load(write_atrium(d))
cam = np.frombuffer(hs.camera.tobytes(), dtype=np.float32)
pos, cdir, fov = cam[0:3].astype(np.float64), cam[3:6].astype(np.float64), float(cam[12])
# generateRay (intersection.cuh:75-97) in float64: good enough for a ray set
u = np.cross(cdir, [0.0, -1.0, 0.0]); u /= np.linalg.norm(u)
v = np.cross(u, cdir); v /= np.linalg.norm(v)
u = -u
dist = (W // 2) / np.tan(fov * 0.5)
xs, ys = np.meshgrid(np.arange(W) - W // 2, np.arange(H) - H // 2)
# 8x8 tiles in row-major tile order, as the megakernel's waves own them
tx, ty = xs.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1), ys.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
dirs = cdir[None, :] * dist + u[None, :] * tx[:, None] + v[None, :] * ty[:, None]
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
primary = np.concatenate([dirs, np.broadcast_to(pos, dirs.shape)], axis=1).astype(np.float32)

faces = hs.faces["vertices"].astype(np.float64)
fn = np.cross(faces[:, 1] - faces[:, 0], faces[:, 2] - faces[:, 0])
fn /= np.maximum(np.linalg.norm(fn, axis=1, keepdims=True), 1e-30)
rng = np.random.default_rng(11)

with P.Context(0) as ctx:
    ctx.setup_function_tables()
    sid = ctx.upload_scene(hs)

    def trace(rays_np, config=0):
        r = torch.from_numpy(rays_np).to(dev)
        o = torch.zeros((len(rays_np), 4), dtype=torch.int32, device=dev)
        ctx.trace_rays_queue(sid, r, o, config)
        torch.cuda.synchronize()
        return o.cpu().numpy()

    sets, rays = [primary], primary
    for gen in range(3):
        rec = trace(rays)
        hit = rec[:, 0] == 1
        t = rec[:, 2].view(np.float32).astype(np.float64)
        o3 = rays[:, 3:6].astype(np.float64) + rays[:, 0:3].astype(np.float64) * t[:, None]
        n = fn[np.clip(rec[:, 1], 0, len(fn) - 1)]
        n = np.where((np.einsum("ij,ij->i", n, rays[:, 0:3].astype(np.float64)) > 0)[:, None], -n, n)
        # cosine-distributed direction about n
        r1, r2 = rng.random(len(rays)), rng.random(len(rays))
        a = np.where(np.abs(n[:, 0:1]) > 0.1, [[0.0, 1.0, 0.0]], [[1.0, 0.0, 0.0]])
        uu = np.cross(a, n); uu /= np.linalg.norm(uu, axis=1, keepdims=True)
        vv = np.cross(n, uu)
        sd = np.sqrt(r1)
        dd = uu * (sd * np.cos(2 * np.pi * r2))[:, None] + vv * (sd * np.sin(2 * np.pi * r2))[:, None] + n * np.sqrt(1 - r1)[:, None]
        nxt = np.concatenate([dd, o3 + dd * 0.03], axis=1).astype(np.float32)[hit]
        sets.append(nxt)
        rays = nxt
    allrays = np.concatenate(sets, axis=0)
    # four "frames": the same rays again with slightly different origins would be fairer to the caches than exact repeats; keep one copy
    n_rays = len(allrays)
    r_dev = torch.from_numpy(allrays).to(dev)
    out = {"scene": "atrium", "rays": n_rays, "rays_by_generation": [len(s) for s in sets], "configs": []}
    ref = None
    names = {0: "16 waves per CU (4 per SIMD), 512-node treelet", 3: "16 waves per CU, 256-node treelet",
             1: "20 waves per CU (5 per SIMD), 2 x 256-node treelets", 2: "24 waves per CU (6 per SIMD), 2 x 256-node treelets"}
    for config in (0, 3, 1, 2):
        for refill in ((8,) if os.environ.get("QUICK") else (8, 1, 16)):
            o_dev = torch.zeros((n_rays, 4), dtype=torch.int32, device=dev)
            waves = ctx.trace_rays_queue(sid, r_dev, o_dev, config, refill)   # warm-up (and the run whose records are compared)
            torch.cuda.synchronize()
            ms = []
            for rep in range(3):   # four launches back to back between two events: host-side launch cost stays out of the figure
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    ctx.trace_rays_queue(sid, r_dev, o_dev, config, refill)
                e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1) / 4.0)
            rec = o_dev.cpu().numpy()
            if ref is None:
                ref = rec
            same = bool(np.array_equal(rec, ref))
            out["configs"].append({"config": config, "what": names[config], "refill_min": refill, "waves_per_cu": waves, "ms": [round(m, 3) for m in ms],
                                   "grays_per_s": round(n_rays / min(ms) / 1e6, 3), "records_identical_to_config_0": same})
    # the records against the one-wave-per-block wide kernel of ptamd_trace_rays on a sample
    idx = rng.choice(n_rays, 20000, replace=False)
    want = ctx.trace_rays(sid, allrays[idx], kernel=P.KERNEL_BVH_RESTART) if hasattr(ctx, "trace_rays") else None
    if want is not None:
        out["sample_equals_ptamd_trace_rays"] = bool(np.array_equal(np.asarray(want), ref[idx]))
    out["hit_share"] = float((ref[:, 0] != 0).mean())
    out["build_id"] = P.native.load().ptamd_build_id().decode()
    print(json.dumps(out, indent=1))
