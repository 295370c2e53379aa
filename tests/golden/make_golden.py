"""Regenerates tests/golden/*.npz.

These fixtures are OUTPUTS OF THE ORACLE (oracle/pt_oracle.c), not of the reference: the
reference cannot be built or run in this environment (needs nvcc + CUDA toolkit + cuRAND)
and ships no golden data.  They pin the oracle against regressions and give the GPU tests a
second, committed expectation.  Inputs are reproducible from the committed assets / seeds.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import cuda_pathtracer_amd as P  # noqa: E402  (loader output is used as input DATA)
import pt_oracle as O  # noqa: E402
from helpers import make_scene, random_soup, synthetic_cubemap  # noqa: E402

CASES = {
    # name: (scene, W, H, spp, bounces, moved, post_id)
    "indoor_64x64_spp2_b3": ("indoor.scene", 64, 64, 2, 3, False, 0),
    "indoor_100x36_spp1_b4_sepia": ("indoor.scene", 100, 36, 1, 4, False, 2),
    "indoor_48x48_moved": ("indoor.scene", 48, 48, 1, 3, True, 0),
    "color_sample_64x48_spp3_b5": ("color_sample.scene", 64, 48, 3, 5, False, 0),
    "island_48x48_spp2_b4": ("island.scene", 48, 48, 2, 4, False, 1),
    "crate_land_56x40_spp2_b3": ("crate_land.scene", 56, 40, 2, 3, False, 3),
}


def textured_scene():
    """Synthetic scene exercising sampleTexture, normal mapping, refraction and a real cubemap."""
    rng = np.random.default_rng(2024)
    tris = random_soup(rng, 96, extent=1.5, size=0.7)
    uvs = rng.uniform(-1.5, 2.5, size=(96, 3, 2)).astype(np.float32)
    textures = [rng.uniform(0.05, 0.95, size=(16, 8, 4)).astype(np.float32),
                rng.uniform(0.0, 1.0, size=(4, 32, 3)).astype(np.float32),
                np.float32([[[0.9, 0.9, 0.9, 0.6]]])]
    materials = [(0, 1, 1.0), (0, -1, 1.0), (2, -1, 1.45), (2, 1, 1.0)]
    mids = rng.integers(0, 4, size=96)
    lights = [((0.5, 2.0, 1.0), (1.0, 0.8, 0.6), 3.0, 0.6), ((-1.5, 0.5, 2.0), (0.5, 0.7, 1.0), 5.0, 0.4)]
    hs = make_scene(P, tris, uvs=uvs, material_ids=mids, materials=materials, textures=textures, lights=lights,
                    mesh_sizes=[40, 56])
    return hs, synthetic_cubemap(rng, 8)


_inputs = {}


def case_inputs(name):
    """(HostScene, cubemap faces) of a case, as the reference would have them on Linux; cached per process."""
    if name not in _inputs:
        if name == "textured_64x64_spp2_b4":
            _inputs[name] = textured_scene()
        else:
            _inputs[name] = _load(CASES[name][0])
    return _inputs[name]


def _load(scene):
    # reference-on-Linux behaviour: textures with '/' paths decode (crate_land: 1024^2 maps + the
    # field_with_house cube cross), indoor.mtl's '\\' paths and cube crosses that are not shipped in
    # assets/ (garden.jpg never existed; water.jpg is not copied) end in the 1x1 fallbacks
    hs = P.HostScene.load(os.path.join(ROOT, "assets", scene))
    return hs, P.cubemap_for_scene(hs, asset_folder=os.path.join(ROOT, "assets"))


def render_case(name):
    hs, cube = case_inputs(name)
    if name == "textured_64x64_spp2_b4":
        W, H, spp, B, moved, post = 64, 64, 2, 4, False, 0
    else:
        _, W, H, spp, B, moved, post = CASES[name]
    acc, rgba = O.render(O.OracleScene.from_host_scene(hs, cube), O.camera_from_record(hs.camera), W, H,
                         spp=spp, bounces=B, moved=moved, post_id=post)
    return hs, cube, dict(W=W, H=H, spp=spp, bounces=B, moved=moved, post_id=post), acc, rgba


ALL = list(CASES) + ["textured_64x64_spp2_b4"]

if __name__ == "__main__":
    for name in ALL:
        _, _, cfg, acc, rgba = render_case(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), accum=acc, rgba=rgba, **cfg)
        print(name, acc.shape, float(acc.sum(dtype=np.float64)))
