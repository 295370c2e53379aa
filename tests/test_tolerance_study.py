"""The stated tolerance w.r.t. the reference's CUDA binary.

The oracle and the HIP kernel agree bit for bit — against each other.  The reference binary is `nvcc -O2` (--fmad=true, libdevice
cosf / sinf / powf, __cosf / __sinf / __fdividef: cuda_opengl/CMakeLists.txt:20-22, raytrace.cu:111-122,163,206,263,
intersection.cuh:113), which nothing here can run.  What CAN be measured is how far faithful builds of this integrator drift apart
when exactly those liberties are taken: scripts/tolerance_study.py builds study instantiations of oracle/pt_oracle.c (fma contraction,
libm's transcendentals, fast-intrinsic stand-ins) and compares their images with the oracle's.  This test holds the numbers that
DESIGN.md states — on a reduced sample here, and on the committed full-size record.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

# THE STATED TOLERANCE (per variant, against the oracle; DESIGN.md section 3 quotes these):
#   headline scene (indoor.obj as the reference loads it: one-texel materials, one-colour environment — a pixel is a function of
#   the SEQUENCE of surfaces its path hits, so it only changes when a hit flips):
INDOOR_MAX_SHARE_BEYOND_1E4 = 1.0e-4      # measured 0 .. 3.3e-5
INDOOR_MAX_SHARE_BEYOND_1LSB = 1.0e-4     # the same pixels
INDOOR_MAX_MEAN_IMAGE_DELTA = 2.0e-6      # measured <= 4.7e-7
#   textured scene (crate_land: radiance is a continuous function of the hit points):
TEXTURED_MAX_SHARE_BEYOND_1E4 = 1.0e-2    # measured 4.6e-4 (libm) .. 6.6e-3 (everything at once); SURVEY 8-c's L1 budget is 5e-3
TEXTURED_MAX_SHARE_BEYOND_1LSB = 1.5e-3   # measured 1e-5 .. 6.1e-4
TEXTURED_MAX_MEAN_IMAGE_DELTA = 1.0e-5    # measured <= 2.5e-6


def check(rec, reduced):
    slack = 3.0 if reduced else 1.0   # a 64-row sample holds a handful of flipped pixels: allow for the small count
    for case in rec["cases"]:
        textured = case["case"].startswith("crate_land")
        for name, r in case["vs_oracle"].items():
            lim = (TEXTURED_MAX_SHARE_BEYOND_1E4, TEXTURED_MAX_SHARE_BEYOND_1LSB, TEXTURED_MAX_MEAN_IMAGE_DELTA) if textured else \
                  (INDOOR_MAX_SHARE_BEYOND_1E4, INDOOR_MAX_SHARE_BEYOND_1LSB, INDOOR_MAX_MEAN_IMAGE_DELTA)
            assert r["share_beyond_1e-4"] <= lim[0] * slack, (case["case"], name, r["share_beyond_1e-4"])
            assert r["share_rgba8_beyond_1lsb"] <= lim[1] * slack, (case["case"], name, r["share_rgba8_beyond_1lsb"])
            assert max(r["mean_image_delta_per_channel"]) <= lim[2] * slack, (case["case"], name, r["mean_image_delta_per_channel"])


def test_committed_record_is_inside_the_stated_tolerance():
    with open(os.path.join(ROOT, "profiles", "r04_tolerance_study.json")) as f:
        rec = json.load(f)
    assert len(rec["cases"]) == 3 and all(len(c["vs_oracle"]) == 5 for c in rec["cases"])
    check(rec, reduced=False)
    # the study instantiations are not the oracle under another name: libm's sincos differs from or_sincosf on a large share of angles
    assert rec["primitives_differ_share"]["libm"]["sincos"] > 0.1
    # ... and contraction does change results where radiance is continuous
    assert rec["cases"][2]["vs_oracle"]["fma"]["identical_pixel_share"] < 0.9


def test_study_instantiations_stay_inside_the_stated_tolerance(tmp_path):
    import tolerance_study as T
    rec = T.run(quick=True, build_dir=str(tmp_path))
    check(rec, reduced=True)
    # configs[0] (256 x 256, 1 spp, 2 bounces) came out bit-identical under every variant when the study was run; hold the claim loosely
    for name, r in rec["cases"][0]["vs_oracle"].items():
        assert r["share_beyond_1e-4"] <= 1e-4, name
