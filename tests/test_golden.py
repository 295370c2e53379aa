"""The oracle against the committed golden fixtures (tests/golden/*.npz — oracle outputs, see
make_golden.py) and size-independent properties of the render contract."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from golden.make_golden import ALL, render_case


@pytest.mark.parametrize("name", ALL)
def test_oracle_reproduces_golden(P, O, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    _, _, cfg, acc, rgba = render_case(name)
    assert (cfg["W"], cfg["H"], cfg["spp"], cfg["bounces"]) == (int(g["W"]), int(g["H"]), int(g["spp"]), int(g["bounces"]))
    np.testing.assert_array_equal(acc.view(np.uint32), g["accum"].view(np.uint32))
    np.testing.assert_array_equal(rgba, g["rgba"])


def test_oracle_is_thread_count_and_band_invariant(P, O, indoor):
    cube = P.cubemap_for_scene(indoor)
    sc, cam = O.OracleScene.from_host_scene(indoor, cube), O.camera_from_record(indoor.camera)
    a1, r1 = O.render(sc, cam, 80, 50, spp=2, bounces=3, nthreads=1)
    a8, r8 = O.render(sc, cam, 80, 50, spp=2, bounces=3, nthreads=8)
    np.testing.assert_array_equal(a1.view(np.uint32), a8.view(np.uint32))
    np.testing.assert_array_equal(r1, r8)
    # rows rendered band by band (the multi-GPU split) == the full frame
    acc = np.zeros((50, 80, 3), np.float32)
    rg = np.zeros((50, 80, 4), np.uint8)
    for k in (1, 2):
        for rows in ((0, 17), (17, 18), (18, 50)):
            _, part = O.render(sc, cam, 80, 50, spp=1, bounces=3, rows=rows, first_frame=k, accum=acc)
            rg[rows[0]:rows[1]] = part[rows[0]:rows[1]]
    np.testing.assert_array_equal(acc.view(np.uint32), a1.view(np.uint32))
    np.testing.assert_array_equal(rg, r1)


def test_accumulation_contract(P, O, indoor):
    """N spp == sum of N clamped 1-spp samples with seeds 1..N (raytrace.cu:255-258); alpha = 0;
    accumulator is row-flipped relative to the surface (raytrace.cu:252 vs :270)."""
    cube = P.cubemap_for_scene(indoor)
    sc, cam = O.OracleScene.from_host_scene(indoor, cube), O.camera_from_record(indoor.camera)
    acc3, rgba = O.render(sc, cam, 40, 24, spp=3, bounces=3)
    parts = [O.render(sc, cam, 40, 24, spp=1, bounces=3, first_frame=k)[0] for k in (1, 2, 3)]
    np.testing.assert_array_equal(acc3, (parts[0] + parts[1]) + parts[2])
    assert (rgba[..., 3] == 0).all()
    assert acc3.min() >= 0.0 and acc3.max() <= 3.0
    # the flip, exactly: surface(y, x) is the tonemapped accumulator entry (H-1-y, x)
    import ctypes as C
    lib = O.load()
    one, rg1 = O.render(sc, cam, 40, 24, spp=1, bounces=3)
    out = (C.c_float * 3)()
    for (y, x) in ((0, 0), (5, 7), (23, 39), (12, 20)):
        a = one[24 - 1 - y, x]
        lib.or_exposure((C.c_float * 3)(*a), out)
        g = float(np.float32(1.0) / np.float32(2.2))
        px = lib.or_pack_rgba((C.c_float * 3)(*[lib.or_powf(out[k], g) for k in range(3)]))
        assert px == int(rg1[y, x].view(np.uint32)[0])
    # moved == preview: one intersect, albedo written straight into a zeroed accumulator (Q11)
    dirty = np.full((24, 40, 3), 5.0, np.float32)
    prev, _ = O.render(sc, cam, 40, 24, spp=1, bounces=3, moved=True, accum=dirty)
    assert prev.max() <= 1.0
