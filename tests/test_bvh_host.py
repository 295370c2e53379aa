"""The acceleration structure's contract, on the CPU: the ordered stackless BVH walk (host
mirror of the device traversal, same records and float ops) returns exactly what the
reference's brute-force loop returns (intersection.cuh:179-196) — same face index, same t bits."""
import os

import numpy as np
import pytest

from conftest import ASSETS
from helpers import make_scene, random_rays, random_soup


def lightless(O, P, hs):
    return O.OracleScene(hs.faces, hs.mesh_sizes, hs.materials, hs.lights[:0], hs.textures, hs.texels,
                         P.cubemap_from_color())


@pytest.mark.parametrize("name", ["indoor", "crate_land", "color_sample", "island", "sss_crate"])
def test_bvh_equals_brute_force_on_assets(P, O, name):
    hs = P.HostScene.load(os.path.join(ASSETS, name + ".scene"))
    rng = np.random.default_rng(7)
    rays = random_rays(rng, 30000, extent=4.0)
    # half of the rays start on surfaces, offset like the path tracer does (raytrace.cu:125-132)
    f = hs.faces["vertices"][rng.integers(0, len(hs.faces), 15000)]
    a, b = rng.uniform(size=(2, 15000, 1)).astype(np.float32)
    flip = (a + b) > 1
    a, b = np.where(flip, 1 - a, a), np.where(flip, 1 - b, b)
    rays[:15000, 3:] = f[:, 0] + a * (f[:, 1] - f[:, 0]) + b * (f[:, 2] - f[:, 0]) + rays[:15000, :3] * np.float32(0.03)
    want = O.intersect(lightless(O, P, hs), rays)
    got, nodes, tris = P.host_bvh_trace(hs, rays)
    np.testing.assert_array_equal(got, want)
    assert (want[:, 0] == 1).sum() > 3000
    assert tris < 0.2 * len(rays) * len(hs.faces)          # it actually culls


def test_bvh_equals_brute_force_on_random_soup(P, O):
    rng = np.random.default_rng(3)
    hs = make_scene(P, random_soup(rng, 700))
    rays = random_rays(rng, 20000)
    np.testing.assert_array_equal(P.host_bvh_trace(hs, rays)[0], O.intersect(lightless(O, P, hs), rays))


def test_ties_resolve_to_lowest_face_index(P, O):
    """Coincident duplicates: the reference's strict `<` keeps the FIRST face in storage order."""
    rng = np.random.default_rng(5)
    base = random_soup(rng, 40)
    tris = np.concatenate([base, base[::-1], base])          # every triangle three times, shuffled positions
    hs = make_scene(P, tris)
    rays = random_rays(rng, 20000)
    want = O.intersect(lightless(O, P, hs), rays)
    got = P.host_bvh_trace(hs, rays)[0]
    np.testing.assert_array_equal(got, want)
    hit = want[want[:, 0] == 1, 1]
    assert len(hit) > 500 and (hit < 40).all()


def test_degenerate_inputs(P, O):
    rng = np.random.default_rng(9)
    rays = random_rays(rng, 2000)
    rays[:50, 0] = 0.0          # axis-parallel directions (inf slab reciprocals)
    rays[50:100, 1:3] = 0.0
    empty = make_scene(P, np.zeros((0, 3, 3), np.float32))
    got = P.host_bvh_trace(empty, rays)[0]
    assert (got[:, 0] == 0).all() and (got[:, 2].view(np.float32) == np.float32(100000.0)).all()
    one = make_scene(P, np.float32([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]]))
    np.testing.assert_array_equal(P.host_bvh_trace(one, rays)[0], O.intersect(lightless(O, P, one), rays))
    # zero-area and NaN triangles never hit and must not poison the tree
    tris = random_soup(rng, 64)
    tris[3] = tris[3][0]
    tris[10, 1, 2] = np.nan
    weird = make_scene(P, tris)
    np.testing.assert_array_equal(P.host_bvh_trace(weird, rays)[0], O.intersect(lightless(O, P, weird), rays))
    # axis-aligned flat geometry: zero-thickness boxes
    quad = np.float32([[[-1, 0, -1], [1, 0, -1], [1, 0, 1]], [[-1, 0, -1], [1, 0, 1], [-1, 0, 1]],
                       [[-1, 0, -1], [1, 0, 1], [1, 0, -1]], [[-1, 0, -1], [-1, 0, 1], [1, 0, 1]]])
    flat = make_scene(P, quad)
    rays[:, 3:] *= 0.5
    np.testing.assert_array_equal(P.host_bvh_trace(flat, rays)[0], O.intersect(lightless(O, P, flat), rays))


def test_reference_presplitting_keeps_the_contract(P, O, tmp_path):
    """PTAMD_BVH_SPLIT_ALPHA (off by default: it did not pay on indoor) represents huge faces
    by several clipped references of the SAME triangle record; results must not change."""
    import subprocess
    import sys
    code = f"""
import sys, numpy as np
sys.path[:0] = [{os.path.dirname(ASSETS)!r}, {os.path.join(os.path.dirname(ASSETS), 'oracle')!r}, {os.path.join(os.path.dirname(ASSETS), 'tests')!r}]
import cuda_pathtracer_amd as P, pt_oracle as O
from helpers import random_rays
hs = P.HostScene.load({os.path.join(ASSETS, 'indoor.scene')!r})
rays = random_rays(np.random.default_rng(5), 20000, 3.4)
sc = O.OracleScene(hs.faces, hs.mesh_sizes, hs.materials, hs.lights[:0], hs.textures, hs.texels, P.cubemap_from_color())
got, nodes, tris = P.host_bvh_trace(hs, rays)
assert (got == O.intersect(sc, rays)).all()
print(nodes, tris)
"""
    base = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.split()
    env = dict(os.environ, PTAMD_TUNING="1", PTAMD_BVH_SPLIT_ALPHA="0.01", PTAMD_BVH_SPLIT_BUDGET="200")
    split = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=env).stdout.split()
    assert base != split          # the knob did change the tree
    # ... and only behind PTAMD_TUNING=1: a production environment cannot change the tree
    ungated = dict(os.environ, PTAMD_BVH_SPLIT_ALPHA="0.01", PTAMD_BVH_SPLIT_BUDGET="200")
    ungated.pop("PTAMD_TUNING", None)
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=ungated).stdout.split() == base


# ------------------------------------------------------------------ the four-wide form (scenes walked from L2)

@pytest.mark.parametrize("name", ["indoor", "crate_land", "color_sample", "island", "sss_crate"])
def test_wide_bvh_equals_brute_force_on_assets(P, O, name):
    hs = P.HostScene.load(os.path.join(ASSETS, name + ".scene"))
    rng = np.random.default_rng(17)
    rays = random_rays(rng, 30000, extent=4.0)
    f = hs.faces["vertices"][rng.integers(0, len(hs.faces), 15000)]
    a, b = rng.uniform(size=(2, 15000, 1)).astype(np.float32)
    flip = (a + b) > 1
    a, b = np.where(flip, 1 - a, a), np.where(flip, 1 - b, b)
    rays[:15000, 3:] = f[:, 0] + a * (f[:, 1] - f[:, 0]) + b * (f[:, 2] - f[:, 0]) + rays[:15000, :3] * np.float32(0.03)
    want = O.intersect(lightless(O, P, hs), rays)
    got, nodes, tris, depth = P.host_bvh4_trace(hs, rays)
    np.testing.assert_array_equal(got, want)
    got2, nodes2, tris2 = P.host_bvh_trace(hs, rays)
    assert nodes < 0.6 * nodes2                     # a wide node visit replaces several box tests of the binary walk
    assert tris < 0.2 * len(rays) * len(hs.faces) and 1 <= depth <= 16


def test_wide_bvh_ties_degenerates_and_soup(P, O):
    rng = np.random.default_rng(23)
    base = random_soup(rng, 40)
    hs = make_scene(P, np.concatenate([base, base[::-1], base]))          # coincident duplicates: lowest index wins
    rays = random_rays(rng, 20000)
    np.testing.assert_array_equal(P.host_bvh4_trace(hs, rays)[0], O.intersect(lightless(O, P, hs), rays))
    soup = make_scene(P, random_soup(rng, 3000, extent=3.0, size=0.25))
    rays = random_rays(rng, 20000)
    got, nodes, tris, depth = P.host_bvh4_trace(soup, rays)
    np.testing.assert_array_equal(got, O.intersect(lightless(O, P, soup), rays))
    rays[:50, 0] = 0.0
    rays[50:100, 1:3] = 0.0
    empty = make_scene(P, np.zeros((0, 3, 3), np.float32))
    got = P.host_bvh4_trace(empty, rays)[0]
    assert (got[:, 0] == 0).all() and (got[:, 2].view(np.float32) == np.float32(100000.0)).all()
    one = make_scene(P, np.float32([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]]))
    np.testing.assert_array_equal(P.host_bvh4_trace(one, rays)[0], O.intersect(lightless(O, P, one), rays))
    tris = random_soup(rng, 64)
    tris[3] = tris[3][0]
    tris[10, 1, 2] = np.nan
    weird = make_scene(P, tris)
    np.testing.assert_array_equal(P.host_bvh4_trace(weird, rays)[0], O.intersect(lightless(O, P, weird), rays))


def test_wide_bvh_on_the_atrium(P, O, tmp_path):
    """configs[3] asset: the wide walk == the binary walk on 20 000 rays (brute force over 264 832 faces is checked on the
    GPU tests' oracle rows), depth of the wide tree bounded."""
    from cuda_pathtracer_amd.synthetic import write_atrium
    hs = P.HostScene.load(write_atrium(str(tmp_path)))
    rng = np.random.default_rng(31)
    rays = random_rays(rng, 20000, extent=5.0)
    rays[:, 3] *= 2.0
    rays[:, 4] = np.abs(rays[:, 4]) + 0.2
    got4, nodes4, tris4, depth = P.host_bvh4_trace(hs, rays)
    got2, nodes2, tris2 = P.host_bvh_trace(hs, rays)
    np.testing.assert_array_equal(got4, got2)
    assert (got4[:, 0] == 1).mean() > 0.5 and depth <= 24
    want = O.intersect(lightless(O, P, hs), rays[:300])
    np.testing.assert_array_equal(got4[:300], want)
    print("atrium: wide nodes/ray %.1f, binary nodes/ray %.1f, depth %d" % (nodes4 / len(rays), nodes2 / len(rays), depth))


# ------------------------------------------------------------------ the eight-wide form with quantised child boxes (round 3)

@pytest.mark.parametrize("name", ["indoor", "crate_land", "color_sample", "island", "sss_crate"])
def test_eight_wide_bvh_equals_brute_force_on_assets(P, O, name):
    """Bvh::nodes8 (origin + per-axis power-of-two scale + 8-bit planes rounded outward, children in direction slots): the
    host mirror of the device walk returns the brute-force record for random rays and for rays launched off the surfaces."""
    hs = P.HostScene.load(os.path.join(ASSETS, name + ".scene"))
    rng = np.random.default_rng(19)
    rays = random_rays(rng, 30000, extent=4.0)
    f = hs.faces["vertices"][rng.integers(0, len(hs.faces), 15000)]
    a, b = rng.uniform(size=(2, 15000, 1)).astype(np.float32)
    flip = (a + b) > 1
    a, b = np.where(flip, 1 - a, a), np.where(flip, 1 - b, b)
    rays[:15000, 3:] = f[:, 0] + a * (f[:, 1] - f[:, 0]) + b * (f[:, 2] - f[:, 0]) + rays[:15000, :3] * np.float32(0.03)
    want = O.intersect(lightless(O, P, hs), rays)
    got, nodes, tris, depth, n_nodes = P.host_bvh8_trace(hs, rays)
    np.testing.assert_array_equal(got, want)
    got4, nodes4, tris4, depth4 = P.host_bvh4_trace(hs, rays)
    assert nodes < 0.75 * nodes4 and 1 <= depth <= 12 and n_nodes >= 1


def test_eight_wide_bvh_ties_degenerates_signed_zeros_and_soup(P, O):
    rng = np.random.default_rng(29)
    base = random_soup(rng, 40)
    hs = make_scene(P, np.concatenate([base, base[::-1], base]))          # coincident duplicates: lowest index wins
    rays = random_rays(rng, 20000)
    np.testing.assert_array_equal(P.host_bvh8_trace(hs, rays)[0], O.intersect(lightless(O, P, hs), rays))
    soup = make_scene(P, random_soup(rng, 3000, extent=3.0, size=0.25))
    rays = random_rays(rng, 20000)
    np.testing.assert_array_equal(P.host_bvh8_trace(soup, rays)[0], O.intersect(lightless(O, P, soup), rays))
    # axis-parallel rays with zeros of either sign (the octant comes from the sign bit; the slab stand-in keeps that sign)
    rays[:50, 0] = 0.0
    rays[50:100, 1:3] = 0.0
    rays[100:150, 0] = -0.0
    rays[150:200, 1:3] = -0.0
    np.testing.assert_array_equal(P.host_bvh8_trace(soup, rays)[0], O.intersect(lightless(O, P, soup), rays))
    empty = make_scene(P, np.zeros((0, 3, 3), np.float32))
    got = P.host_bvh8_trace(empty, rays)[0]
    assert (got[:, 0] == 0).all() and (got[:, 2].view(np.float32) == np.float32(100000.0)).all()
    one = make_scene(P, np.float32([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]]))
    np.testing.assert_array_equal(P.host_bvh8_trace(one, rays)[0], O.intersect(lightless(O, P, one), rays))
    tris = random_soup(rng, 64)
    tris[3] = tris[3][0]
    tris[10, 1, 2] = np.nan
    weird = make_scene(P, tris)
    np.testing.assert_array_equal(P.host_bvh8_trace(weird, rays)[0], O.intersect(lightless(O, P, weird), rays))
    # huge and tiny scenes: the per-axis exponent follows the extent
    for scale in (1.0e-4, 3.0e5):
        big = make_scene(P, random_soup(rng, 500, extent=3.0, size=0.4) * np.float32(scale))
        r = random_rays(rng, 4000, extent=3.0)
        r[:, 3:] *= np.float32(scale)
        np.testing.assert_array_equal(P.host_bvh8_trace(big, r)[0], O.intersect(lightless(O, P, big), r))


def test_eight_wide_bvh_on_the_atrium(P, O, tmp_path):
    """configs[3] asset: the eight-wide walk == the binary walk on 20 000 rays, at most 10 node visits per ray, two thirds
    of them in the part of the tree the device keeps in LDS."""
    from cuda_pathtracer_amd.synthetic import write_atrium
    hs = P.HostScene.load(write_atrium(str(tmp_path)))
    rng = np.random.default_rng(31)
    rays = random_rays(rng, 20000, extent=5.0)
    rays[:, 3] *= 2.0
    rays[:, 4] = np.abs(rays[:, 4]) + 0.2
    got8, nodes8, tris8, depth, n_nodes = P.host_bvh8_trace(hs, rays)
    top73, top585 = P.host_bvh8_trace.top_visits
    got2, nodes2, tris2 = P.host_bvh_trace(hs, rays)
    np.testing.assert_array_equal(got8, got2)
    want = O.intersect(lightless(O, P, hs), rays[:300])
    np.testing.assert_array_equal(got8[:300], want)
    assert nodes8 / len(rays) <= 10.0 and depth <= 12 and top585 >= 0.6 * nodes8
    print("atrium: eight-wide nodes/ray %.2f (%.2f in the first 585 nodes), triangles/ray %.2f, depth %d, %d nodes; binary nodes/ray %.1f"
          % (nodes8 / len(rays), top585 / len(rays), tris8 / len(rays), depth, n_nodes, nodes2 / len(rays)))


# ------------------------------------------------------------------ the four-wide form in 64-byte quantised nodes (round 3)

@pytest.mark.parametrize("name", ["indoor", "crate_land", "color_sample", "island", "sss_crate"])
def test_quantised_four_wide_bvh_equals_brute_force_on_assets(P, O, name):
    """Bvh::nodes4q (origin + per-axis power-of-two scale + 8-bit planes rounded outward, order stored for four octants and
    read inverted for the opposite four): the host mirror of the device walk returns the brute-force record for random rays
    and for rays launched off the surfaces, and visits only slightly more nodes than the float form (looser boxes)."""
    hs = P.HostScene.load(os.path.join(ASSETS, name + ".scene"))
    rng = np.random.default_rng(23)
    rays = random_rays(rng, 30000, extent=4.0)
    f = hs.faces["vertices"][rng.integers(0, len(hs.faces), 15000)]
    a, b = rng.uniform(size=(2, 15000, 1)).astype(np.float32)
    flip = (a + b) > 1
    a, b = np.where(flip, 1 - a, a), np.where(flip, 1 - b, b)
    rays[:15000, 3:] = f[:, 0] + a * (f[:, 1] - f[:, 0]) + b * (f[:, 2] - f[:, 0]) + rays[:15000, :3] * np.float32(0.03)
    want = O.intersect(lightless(O, P, hs), rays)
    got, nodes, tris, depth = P.host_bvh4q_trace(hs, rays)
    np.testing.assert_array_equal(got, want)
    assert 1 <= depth <= 24 and nodes > 0


def test_quantised_four_wide_bvh_ties_degenerates_signed_zeros_and_soup(P, O):
    rng = np.random.default_rng(37)
    base = random_soup(rng, 40)
    hs = make_scene(P, np.concatenate([base, base[::-1], base]))          # coincident duplicates: lowest index wins
    rays = random_rays(rng, 20000)
    np.testing.assert_array_equal(P.host_bvh4q_trace(hs, rays)[0], O.intersect(lightless(O, P, hs), rays))
    soup = make_scene(P, random_soup(rng, 3000, extent=3.0, size=0.25))
    rays = random_rays(rng, 20000)
    np.testing.assert_array_equal(P.host_bvh4q_trace(soup, rays)[0], O.intersect(lightless(O, P, soup), rays))
    # axis-parallel rays with zeros of either sign (the planes a ray enters through follow the SIGN BIT of its direction)
    rays[:50, 0] = 0.0
    rays[50:100, 1:3] = 0.0
    rays[100:150, 0] = -0.0
    rays[150:200, 1:3] = -0.0
    np.testing.assert_array_equal(P.host_bvh4q_trace(soup, rays)[0], O.intersect(lightless(O, P, soup), rays))
    empty = make_scene(P, np.zeros((0, 3, 3), np.float32))
    got = P.host_bvh4q_trace(empty, rays)[0]
    assert (got[:, 0] == 0).all() and (got[:, 2].view(np.float32) == np.float32(100000.0)).all()
    one = make_scene(P, np.float32([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]]))
    np.testing.assert_array_equal(P.host_bvh4q_trace(one, rays)[0], O.intersect(lightless(O, P, one), rays))
    tris = random_soup(rng, 64)
    tris[3] = tris[3][0]
    tris[10, 1, 2] = np.nan
    weird = make_scene(P, tris)
    np.testing.assert_array_equal(P.host_bvh4q_trace(weird, rays)[0], O.intersect(lightless(O, P, weird), rays))
    for scale in (1.0e-4, 3.0e5):
        big = make_scene(P, random_soup(rng, 500, extent=3.0, size=0.4) * np.float32(scale))
        r = random_rays(rng, 4000, extent=3.0)
        r[:, 3:] *= np.float32(scale)
        np.testing.assert_array_equal(P.host_bvh4q_trace(big, r)[0], O.intersect(lightless(O, P, big), r))


def test_quantised_four_wide_bvh_on_the_atrium(P, O, tmp_path):
    """configs[3] asset: the quantised walk == the binary walk on 20 000 rays; node visits within 15 % of the float form."""
    from cuda_pathtracer_amd.synthetic import write_atrium
    hs = P.HostScene.load(write_atrium(str(tmp_path)))
    rng = np.random.default_rng(31)
    rays = random_rays(rng, 20000, extent=5.0)
    rays[:, 3] *= 2.0
    rays[:, 4] = np.abs(rays[:, 4]) + 0.2
    gotq, nodesq, trisq, depth = P.host_bvh4q_trace(hs, rays)
    got2, nodes2, tris2 = P.host_bvh_trace(hs, rays)
    np.testing.assert_array_equal(gotq, got2)
    got4, nodes4, tris4, depth4 = P.host_bvh4_trace(hs, rays)
    assert nodesq <= 1.15 * nodes4
    print("atrium: quantised four-wide nodes/ray %.2f (float form %.2f), triangles/ray %.2f (%.2f)"
          % (nodesq / len(rays), nodes4 / len(rays), trisq / len(rays), tris4 / len(rays)))


def test_infinite_and_huge_coordinates(P, O):
    """A coordinate that is infinite (either sign), NaN, or near the end of the float range: the builder keeps non-finite
    coordinates out of the boxes (such a face can never be hit), every walk returns the brute-force record; the quantised forms
    are for coordinates within +-1e8 (the library walks the float nodes beyond)."""
    rng = np.random.default_rng(83)
    base = random_soup(rng, 80, extent=1.2, size=0.9)
    rays = random_rays(rng, 6000, extent=2.0)
    rays[:30, 0] = 0.0
    rays[30:60, 1] = -0.0
    for bad in (np.inf, -np.inf, np.nan, 3.0e38, -3.0e38, 3.0e9, -9.0e7):
        tris = base.copy()
        tris[7, 1, 0] = np.float32(bad)
        tris[11, 2, 2] = np.float32(bad)
        hs = make_scene(P, tris)
        want = O.intersect(lightless(O, P, hs), rays)
        np.testing.assert_array_equal(P.host_bvh_trace(hs, rays)[0], want)
        np.testing.assert_array_equal(P.host_bvh4_trace(hs, rays)[0], want)
        if not np.isfinite(bad) or abs(bad) <= 1.0e8:
            np.testing.assert_array_equal(P.host_bvh4q_trace(hs, rays)[0], want)
            np.testing.assert_array_equal(P.host_bvh8_trace(hs, rays)[0], want)
