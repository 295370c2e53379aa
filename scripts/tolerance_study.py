#!/usr/bin/env python3
"""How far can two FAITHFUL builds of this integrator drift apart?  (CPU only)

The reference binary is `nvcc -O2` and nothing else (cuda_opengl/CMakeLists.txt:20-22): --fmad=true, libdevice's cosf / sinf /
powf / tanf and the fast intrinsics __cosf / __sinf / __fdividef (raytrace.cu:111-122,163,206,263, intersection.cuh:79,113).
The oracle (and the HIP kernel, bit for bit) forbids contraction and uses its own sincos / powf.  Nothing reference-held pins
either, so the honest statement is the SIZE of the gap between such builds: this script renders BASELINE configs[0] and a
256-row crop of configs[1] with the oracle and with study instantiations of the same source (oracle/pt_oracle.c: OR_STUDY_*)
and reports, per variant, against the oracle:

  * share of pixels whose mean-image value (accumulator / spp) differs by more than 1e-4 in any channel,
  * share of pixels whose RGBA8 differs by more than +-1 LSB in any channel, and at all,
  * the histogram of per-pixel max-channel |delta|, the largest |delta|,
  * the mean-image delta: |mean(V) - mean(O)| per channel, and the mean absolute delta.

    python3 scripts/tolerance_study.py [--out profiles/r04_tolerance_study.json] [--quick]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("PTAMD_NO_TORCH_PRELOAD", "1")

VARIANTS = {
    "fma": ["-ffp-contract=fast"],
    "libm": ["-DOR_STUDY_LIBM"],
    "fma+libm": ["-ffp-contract=fast", "-DOR_STUDY_LIBM"],
    "fastintr": ["-DOR_STUDY_FASTINTR"],
    "fma+libm+fastintr": ["-ffp-contract=fast", "-DOR_STUDY_LIBM", "-DOR_STUDY_FASTINTR"],
}
EDGES = [0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0 + 1e-9]


def compare(acc_o, rgba_o, acc_v, rgba_v, spp):
    mo, mv = acc_o / np.float32(spp), acc_v / np.float32(spp)
    d = np.abs(mv.astype(np.float64) - mo.astype(np.float64))
    dmax = np.nan_to_num(d, nan=1.0).max(axis=2)
    n = dmax.size
    db = np.abs(rgba_v[..., :3].astype(np.int32) - rgba_o[..., :3].astype(np.int32)).max(axis=2)
    hist, _ = np.histogram(dmax, bins=EDGES)
    return {
        "pixels": int(n),
        "identical_pixel_share": float((dmax == 0.0).sum() / n),
        "share_beyond_1e-4": float((dmax > 1e-4).sum() / n),
        "share_rgba8_beyond_1lsb": float((db > 1).sum() / n),
        "share_rgba8_differs": float((db > 0).sum() / n),
        "max_abs_delta": float(dmax.max()),
        "hist_edges": EDGES[:-1] + [1.0],
        "hist_counts": [int(v) for v in hist],
        "mean_image_delta_per_channel": [float(abs(mv[..., c].astype(np.float64).mean() - mo[..., c].astype(np.float64).mean())) for c in range(3)],
        "mean_abs_delta": float(d.mean()),
    }


def run(quick=False, build_dir=None, nthreads=0):
    import cuda_pathtracer_amd as P
    import pt_oracle as O
    build_dir = build_dir or os.path.join(ROOT, "build", "oracle_study")
    libs = {name: O.build_variant(os.path.join(build_dir, "libpt_oracle_" + name.replace("+", "_") + ".so"), flags) for name, flags in VARIANTS.items()}
    scenes = {}
    for sname in ("indoor", "crate_land"):
        hs = P.HostScene.load(os.path.join(ROOT, "assets", sname + ".scene"))
        cube = P.cubemap_for_scene(hs, asset_folder=os.path.join(ROOT, "assets"))
        scenes[sname] = (O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), hs, cube)
    # configs[0]: 256x256, 1 spp, 2 bounces; configs[1] crop: 256 rows around the middle of 1920x1080, 4 spp, 4 bounces;
    # and a scene whose radiance is a CONTINUOUS function of the hit points (1024^2 textures, normal maps, bilinear cubemap):
    # on indoor.obj (one-texel materials, one-colour environment) a pixel only changes when a hit flips to another surface
    cases = [("configs[0]: indoor 256x256, 1 spp, 2 bounces", "indoor", 256, 256, 1, 2, (0, 256)),
             ("configs[1] crop: indoor 1920x1080, rows [412, 668), 4 spp, 4 bounces", "indoor", 1920, 1080, 4, 4, (412, 668) if not quick else (508, 572)),
             ("crate_land (1024^2 RGBA + normal maps, bilinear cubemap) 480x270, 4 spp, 4 bounces", "crate_land", 480, 270, 4, 4, (0, 270) if not quick else (100, 164))]
    out = {"what": __doc__.split("\n")[0], "variants": {k: " ".join(v) for k, v in VARIANTS.items()}, "cases": []}
    # do the instantiations differ at all?  share of 20 000 angles in (0, 2 pi] / bases in (0, 1] whose sincos / pow(x, 5) / pow(x, 1 / 2.2) bits differ
    import ctypes as C
    rs = np.random.RandomState(1234)
    ang = (rs.rand(20000) * 2 * np.pi).astype(np.float32)
    xs = rs.rand(20000).astype(np.float32)
    def prim(lib):
        s_, c_ = C.c_float(), C.c_float()
        sc = []
        for a in ang:
            lib.or_sincosf(C.c_float(float(a)), C.byref(s_), C.byref(c_)); sc.append((s_.value, c_.value))
        return (np.array(sc, dtype=np.float32).view(np.uint32), np.array([lib.or_powf(float(x), 5.0) for x in xs], dtype=np.float32).view(np.uint32),
                np.array([lib.or_powf(float(x), 1.0 / 2.2) for x in xs], dtype=np.float32).view(np.uint32))
    base = prim(O.load())
    out["primitives_differ_share"] = {}
    for vname, lib in libs.items():
        v = prim(lib)
        out["primitives_differ_share"][vname] = {"sincos": float((v[0] != base[0]).any(axis=1).mean()), "pow5": float((v[1] != base[1]).mean()),
                                                 "pow_1_2.2": float((v[2] != base[2]).mean())}
    for name, sname, w, h, spp, b, rows in cases:
        osc, cam, hs, cube = scenes[sname]
        acc_o, rgba_o = O.render(osc, cam, w, h, spp=spp, bounces=b, rows=rows, nthreads=nthreads)
        crop = slice(rows[0], rows[1])
        acrop = slice(h - rows[1], h - rows[0])   # the accumulator is row-flipped (raytrace.cu:252)
        rec = {"case": name, "rows": list(rows), "vs_oracle": {}}
        for vname, lib in libs.items():
            acc_v, rgba_v = O.render(osc, cam, w, h, spp=spp, bounces=b, rows=rows, nthreads=nthreads, lib=lib)
            rec["vs_oracle"][vname] = compare(acc_o[acrop], rgba_o[crop], acc_v[acrop], rgba_v[crop], spp)
        out["cases"].append(rec)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_tolerance_study.json"))
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    res = run(a.quick)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print("primitives whose bits differ from the oracle's:", res["primitives_differ_share"])
    for c in res["cases"]:
        print(c["case"])
        for v, r in c["vs_oracle"].items():
            print("  %-20s identical %.4f  >1e-4 %.5f  rgba8 >1 LSB %.5f  rgba8 differs %.5f  max %.3g  mean-image delta %.2e  mean |delta| %.2e" % (
                v, r["identical_pixel_share"], r["share_beyond_1e-4"], r["share_rgba8_beyond_1lsb"], r["share_rgba8_differs"],
                r["max_abs_delta"], max(r["mean_image_delta_per_channel"]), r["mean_abs_delta"]))
