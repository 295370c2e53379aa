#!/usr/bin/env python3
"""Single-GPU proxy of the N-GPU row-band split (configs[2]): renders each of the N bands of the 1920x1080 frame on
ONE MI355X, exactly as rank r of an N-rank job would (band-local buffers, batched launch, k frames in flight on 1/k-GPU
launches, the RGBA8 band all-gathered through RCCL — a 1-rank communicator here, so the xGMI transfer is NOT in it), and
records ms per frame per band.  max(band) / mean(band) is the load imbalance a contiguous split would see; SURVEY §8-e
prescribes interleaved bands above 5 %.

    python scripts/band_proxy.py [--ranks 8] [--out profiles/r02_band_proxy.json]
"""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--ranks", type=int, nargs="+", default=[8, 4, 2])
ap.add_argument("--in-flight", type=int, nargs="+", default=[1, 3, 4])
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--machine-share", type=int, default=0, help="passed to bench.py (0: = frames in flight)")
ap.add_argument("--interleave", type=int, default=0, help="interleaved bands of this many rows instead of contiguous bands")
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_band_proxy.json"))
args = ap.parse_args()
sys.path.insert(0, ROOT)
import cuda_pathtracer_amd as P  # noqa: E402  (row_bands only; no GPU use in this process)

H = 1080
res = {"frame": "indoor.scene 1920x1080 4 spp 4 bounces", "kernel": "restart (default)",
       "assignment": f"interleaved {args.interleave}-row bands (band j -> rank j % N)" if args.interleave else "contiguous bands", "gather": "RCCL all-gather forced (1-rank communicator)",
       "steps": args.steps, "splits": []}
for n in args.ranks:
    for fif in args.in_flight:
        bands = []
        for r, (y0, y1) in enumerate(P.row_bands(H, n)):
            env = dict(os.environ, PTAMD_BENCH_FORCE_GATHER="1")
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--as-rank", f"{r}/{n}", "--interleave", str(args.interleave),
                                  "--steps", str(args.steps), "--warmup", "5", "--no-cpu-baseline", "--no-extra",
                                  "--frames-in-flight", str(fif), "--machine-share", str(args.machine_share)], env=env, capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                print(out.stderr[-2000:], file=sys.stderr)
                raise SystemExit(f"bench failed for rank {r}/{n}")
            d = json.loads(out.stdout.strip().splitlines()[-1])
            bands.append({"rank": r, "rows": [y0, y1] if not args.interleave else f"bands {r}, {r + n}, ... of {args.interleave} rows",
                          "ms_per_frame": d["ms_per_step"], "kernel_ms_per_launch": d["roofline"]["kernel_ms_per_launch"]})
        ms = [b["ms_per_frame"] for b in bands]
        mean = sum(ms) / len(ms)
        entry = {"ranks": n, "frames_in_flight": fif, "bands": bands, "max_ms": max(ms), "mean_ms": round(mean, 4),
                 "max_over_mean": round(max(ms) / mean, 4),
                 # an N-GPU frame takes as long as its slowest band
                 "implied_msamples_per_s": round(1920 * 1080 * 4 / (max(ms) * 1e-3) / 1e6, 1)}
        res["splits"].append(entry)
        print(f"ranks {n} in flight {fif}: max {max(ms):.4f} mean {mean:.4f} max/mean {max(ms) / mean:.3f} -> {entry['implied_msamples_per_s']} Msamples/s", flush=True)
with open(args.out, "w") as f:
    json.dump(res, f, indent=1)
