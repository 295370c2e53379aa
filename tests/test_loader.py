"""Host loader (.scene / OBJ / MTL -> flattened C-ABI arrays) against the reference's behaviour.

Pins: the counts the survey observed when the reference loader ran on these assets
(indoor 10/446/6/3, crate_land 13/146/2, color_sample 8/122/6 with ior 1.5,
island 5/194/2 with ior 1.33), the .scene grammar (scene.cpp:61-155) and an independent
pure-Python OBJ reader for the geometry.
"""
import math
import os

import numpy as np
import pytest

from conftest import ASSETS


def scene(P, name):
    return P.HostScene.load(os.path.join(ASSETS, name))


@pytest.mark.parametrize("name,meshes,faces,mats,lights", [
    ("indoor.scene", 10, 446, 6, 3),
    ("crate_land.scene", 13, 146, 2, 1),
    ("color_sample.scene", 8, 122, 6, 0),
    ("island.scene", 5, 194, 2, 0),
    ("sss_crate.scene", None, None, None, 1),
])
def test_scene_counts(P, name, meshes, faces, mats, lights):
    hs = scene(P, name)
    if meshes is not None:
        assert len(hs.mesh_sizes) == meshes
        assert len(hs.faces) == faces == int(hs.mesh_sizes.sum())
        assert len(hs.materials) == mats
    assert len(hs.lights) == lights
    assert (hs.faces["material_id"] < len(hs.materials)).all()


def test_indoor_camera_and_lights(indoor):
    cam = indoor.camera
    np.testing.assert_array_equal(cam["position"], np.float32([-2.7, 2.06, 2.52]))
    d = np.float32([0.62, -0.348, -0.7])
    inv = np.float32(1.0) / np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))  # normalize (cutils_math.h:1557)
    np.testing.assert_array_equal(cam["dir"], d * inv)
    assert cam["fov_x"] == np.float32(90.0 * math.pi / 180.0)     # scene.cpp:76
    assert cam["focus_dist"] == np.float32(3.555) and cam["aperture"] == np.float32(0.01) and cam["speed"] == np.float32(1.4)
    np.testing.assert_array_equal(indoor.lights["vec"], np.float32([[2.9, 2.1, 2.9], [-2.7, 2.1, -2.55], [2.9, 2.1, -2.55]]))
    assert (indoor.lights["emission"] == 2.0).all() and (indoor.lights["radius"] == np.float32(0.3)).all()
    assert (indoor.lights["color"] == 1.0).all()
    assert indoor.cubemap == "cubemap/garden.jpg"


def test_indoor_materials_degrade_to_unit_textures(indoor):
    """indoor.mtl uses backslash texture paths: on Linux every load fails (material_loader.cpp:97-104)
    -> 1x1 RGBA of (Kd, mean Ks), no normal maps, ior 1."""
    expect = [  # floor frame lamp screen wall wood  (MTL order = material ids)
        (0.64, 0.5), (0.063588, 0.09375), (0.310813, 0.177083), (0.334323, 1.0), (1.0, 0.0), (0.8, 0.0)]
    assert len(indoor.textures) == 6
    for i, (kd, ks) in enumerate(expect):
        m = indoor.materials[i]
        assert m["diffuse_spec_map"] == i and m["normal_map"] == -1 and m["ior"] == 1.0
        t = indoor.textures[i]
        assert (t["w"], t["h"], t["nb_chan"]) == (1, 1, 4)
        px = indoor.texels[t["offset"]: t["offset"] + 4]
        ks32 = np.float32(ks)
        mean_ks = np.float32(np.float64(ks32 + ks32 + ks32) / 3.0)   # material_loader.cpp:174-176
        np.testing.assert_array_equal(px, np.float32([kd, kd, kd, mean_ks]))


def test_iors(P):
    cs = scene(P, "color_sample.scene")
    assert cs.materials["ior"][4] == np.float32(1.5)
    isl = scene(P, "island.scene")
    assert np.float32(1.33) in isl.materials["ior"]


def py_obj_triangles(path):
    """Independent OBJ reader: positions of fan-triangulated faces in file order."""
    v, tris = [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                v.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                idx = [int(tok.split("/")[0]) for tok in p[1:]]
                idx = [i - 1 if i > 0 else len(v) + i for i in idx]
                for k in range(2, len(idx)):
                    tris.append([v[idx[0]], v[idx[k - 1]], v[idx[k]]])
    return np.asarray(tris, dtype=np.float64).astype(np.float32)


@pytest.mark.parametrize("name", ["indoor", "crate_land", "color_sample", "island", "sss_crate"])
def test_geometry_matches_independent_obj_reader(P, name):
    hs = scene(P, name + ".scene")
    ref = py_obj_triangles(os.path.join(ASSETS, "obj", name + ".obj"))
    assert hs.faces["vertices"].shape == ref.shape
    np.testing.assert_array_equal(hs.faces["vertices"], ref)


def test_tangent_formula(indoor):
    """scene.cpp:251-261 (including the NaN/inf tangents of faces with degenerate UVs)."""
    f = indoor.faces
    with np.errstate(all="ignore"):
        e1 = f["vertices"][:, 1] - f["vertices"][:, 0]
        e2 = f["vertices"][:, 2] - f["vertices"][:, 0]
        d1 = f["texcoords"][:, 1] - f["texcoords"][:, 0]
        d2 = f["texcoords"][:, 2] - f["texcoords"][:, 0]
        ff = np.float32(1.0) / (d1[:, 0] * d2[:, 1] - d2[:, 0] * d1[:, 1])
        tan = ff[:, None] * (d2[:, 1:2] * e1 - d1[:, 1:2] * e2)
    np.testing.assert_array_equal(f["tangent"].view(np.uint32), tan.astype(np.float32).view(np.uint32))
    assert (~np.isfinite(f["tangent"]).all(axis=1)).sum() == 8


def test_scene_grammar_defaults_and_errors(P, tmp_path):
    obj = tmp_path / "t.obj"
    obj.write_text("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 0 1\n"
                   "usemtl a\nf 1/1/1 2/2/1 4/3/1 3/3/1\ng second\nusemtl b\nf -4/1/1 -3/2/1 -1/3/1\n")
    (tmp_path / "t.mtl").write_text("newmtl a\nKd 0.1 0.2 0.3\nKs 0.3 0.6 0.9\nNi 1.5\n\nnewmtl b\nKd 1 1 1\n")
    sc = tmp_path / "t.scene"
    sc.write_text("# comment\ncamera 1 2 3 0 0 -2 60\np_light 1 2 3 0.5 0.25 1 7 0.5\np_light 1 2\nscene t.obj\ncubemap 0xff0000\n")
    hs = P.HostScene.load(str(sc))
    assert list(hs.mesh_sizes) == [2, 1]                    # quad fanned into 2, `g` starts a new mesh
    np.testing.assert_array_equal(hs.faces["vertices"][1], np.float32([[0, 0, 0], [1, 1, 0], [0, 1, 0]]))  # fan (0, k-1, k)
    np.testing.assert_array_equal(hs.faces["vertices"][2], np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0]]))  # negative indices
    assert list(hs.faces["material_id"]) == [0, 0, 1]
    assert hs.materials["ior"][0] == np.float32(1.5) and hs.materials["ior"][1] == 1.0
    np.testing.assert_array_equal(hs.texels[:4], np.float32([0.1, 0.2, 0.3, np.float32((np.float32(0.3) + np.float32(0.6) + np.float32(0.9)) / 3.0)]))
    cam = hs.camera
    np.testing.assert_array_equal(cam["dir"], np.float32([0, 0, -1]))
    assert cam["fov_x"] == np.float32(60.0 * math.pi / 180.0)
    assert cam["focus_dist"] == 2.0 and cam["aperture"] == 0.125     # scene.cpp:78-79 defaults
    assert len(hs.lights) == 1                                       # the truncated p_light line is skipped
    assert hs.lights[0]["emission"] == 7.0 and hs.lights[0]["radius"] == 0.5
    assert hs.cubemap == "0xff0000"
    with pytest.raises(P.PtamdError) as e:
        P.HostScene.load(str(tmp_path / "missing.scene"))
    assert e.value.status == P.native.PTAMD_ERR_IO
    bad = tmp_path / "bad.scene"
    bad.write_text("scene nothere.obj\n")
    with pytest.raises(P.PtamdError):
        P.HostScene.load(str(bad))
    nomtl = tmp_path / "nomtl.obj"
    nomtl.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    (tmp_path / "nomtl.scene").write_text("scene nomtl.obj\n")
    with pytest.raises(P.PtamdError):      # reference would index materials[-1]; we refuse
        P.HostScene.load(str(tmp_path / "nomtl.scene"))


def test_cubemap_helpers(P):
    c = P.cubemap_from_color(0x131B23)
    assert c.shape == (6, 1, 1, 4)
    np.testing.assert_array_equal(c[3, 0, 0], np.float32([19 / 255, 27 / 255, 35 / 255, 0]).astype(np.float32))
    # cube cross 4x3 of 2x2 faces: value encodes (row, col) of the source texel
    size = 2
    cross = np.zeros((3 * size, 4 * size, 3), np.float32)
    for y in range(3 * size):
        for x in range(4 * size):
            cross[y, x] = (y, x, 7)
    faces = P.cubemap_from_cross(cross)
    assert faces.shape == (6, 2, 2, 4)
    # +x=(col2,row1) -x=(col0,row1) +y=(col1,row0) -y=(col1,row2) +z=(col1,row1) -z=(col3,row1) (texture_utils.cpp:27-52)
    for f, (col, row) in enumerate([(2, 1), (0, 1), (1, 0), (1, 2), (1, 1), (3, 1)]):
        np.testing.assert_array_equal(faces[f, 0, 0], np.float32([row * size, col * size, 7, 0]))
        np.testing.assert_array_equal(faces[f, 1, 1], np.float32([row * size + 1, col * size + 1, 7, 0]))
    with pytest.raises(P.PtamdError):
        P.cubemap_from_cross(np.zeros((9, 12, 3), np.float32))      # size 3 is not a power of two
    with pytest.raises(P.PtamdError):
        P.cubemap_from_cross(np.zeros((4, 8, 3), np.float32))       # not a 4x3 cross
    # the reference always ends at the default colour when the image cannot be loaded
    hs = P.HostScene.load(os.path.join(ASSETS, "indoor.scene"))
    np.testing.assert_array_equal(P.cubemap_for_scene(hs), c)


def test_ppm_roundtrip(P, tmp_path):
    img = (np.arange(5 * 7 * 4) % 256).astype(np.uint8).reshape(5, 7, 4)
    P.save_ppm(str(tmp_path / "x.ppm"), img)
    np.testing.assert_array_equal(P.load_ppm(str(tmp_path / "x.ppm")), img[:, :, :3])
    with pytest.raises(ValueError):
        P.save_ppm(str(tmp_path / "y.ppm"), img.astype(np.float32))


def test_png_writer_roundtrip(P, tmp_path):
    """ptamd_image_save_png -> the built-in PNG reader and PIL both give the pixels back; 4-channel surfaces drop alpha by default."""
    from PIL import Image
    rng = np.random.default_rng(0)
    path = str(tmp_path / "w.png")
    for shape in ((1, 1, 3), (37, 53, 4), (5, 7, 1), (9, 3, 2), (300, 250, 3)):          # the last one spans several stored blocks
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        P.save_png(path, a, drop_alpha=False)
        np.testing.assert_array_equal(P.load_image8(path), a)
        with Image.open(path) as im:
            im.load()
            np.testing.assert_array_equal(np.asarray(im).reshape(shape), a)
    rgba = rng.integers(0, 256, size=(4, 6, 4), dtype=np.uint8)
    P.save_png(path, rgba)
    np.testing.assert_array_equal(P.load_image8(path), rgba[:, :, :3])
    with pytest.raises(ValueError):
        P.save_png(path, rgba.astype(np.float32))
    with pytest.raises(P.native.PtamdError):
        P.save_png(str(tmp_path / "no_such_dir" / "x.png"), rgba)


def _fake_images():
    """A tiny image set served through the provider callback: name -> float32[h, w, c]."""
    rng = np.random.default_rng(1)
    return {
        "rgb4x2.img": rng.uniform(0, 1, (2, 4, 3)).astype(np.float32),
        "a4x2.img": rng.uniform(0, 1, (2, 4, 1)).astype(np.float32),
        "a2x1.img": rng.uniform(0, 1, (1, 2, 1)).astype(np.float32),
        "nrm3x3.img": rng.uniform(0, 1, (3, 3, 3)).astype(np.float32),
        "gray.img": rng.uniform(0, 1, (2, 2, 1)).astype(np.float32),
        "sub/dir.img": rng.uniform(0, 1, (1, 1, 3)).astype(np.float32),
    }


def test_material_packing_rules_with_image_provider(P, tmp_path):
    """material_loader.cpp:243-401 through an injected image provider (the role of stbi_loadf)."""
    imgs = _fake_images()
    calls = []

    def provider(path):
        calls.append(path)
        for k, v in imgs.items():
            if path.endswith("/" + k):
                return v
        return None

    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\n" +
                                    "".join(f"usemtl m{i}\nf 1/1/1 2/1/1 3/1/1\n" for i in range(9)))
    (tmp_path / "m.mtl").write_text(
        "newmtl m0\nKd 0.1 0.2 0.3\nKs 0.3 0.3 0.3\n"                                        # case 1: no maps
        "newmtl m1\nKd 0.1 0.2 0.3\nKs 0.6 0.6 0.6\nmap_Kd rgb4x2.img\n"                     # case 2: rgb only
        "newmtl m2\nKd 0.4 0.5 0.6\nKs 0.9 0.9 0.9\nmap_Ks a4x2.img\n"                       # case 2: a only
        "newmtl m3\nKd 0.4 0.5 0.6\nKs 0.0 0.0 0.0\nmap_Kd rgb4x2.img\nmap_Ks a4x2.img\nmap_Bump nrm3x3.img\n"  # case 3
        "newmtl m4\nKd 0.7 0.7 0.7\nKs 0.0 0.0 0.0\nmap_Kd rgb4x2.img\nmap_Ks a4x2.img\nnorm nrm3x3.img\n"      # cached pack + cached normal
        "newmtl m5\nKd 0.7 0.8 0.9\nKs 0.3 0.3 0.3\nmap_Kd missing.img\nmap_Bump gray.img\n"  # rgb fails -> unit; 1-channel normal dropped
        "newmtl m6\nKd 0.2 0.2 0.2\nKs 0.5 0.5 0.5\nmap_Kd missing.img\nmap_Ks a4x2.img\n"   # case 3, rgb failed: pack(a, default rgb)
        "newmtl m7\nKd 0.2 0.2 0.2\nKs 0.5 0.5 0.5\nmap_Kd rgb4x2.img\nmap_Ks a2x1.img\n"    # sizes differ: smaller map resampled
        "newmtl m8\nKd 0.2 0.2 0.2\nKs 0.5 0.5 0.5\nmap_Kd sub\\dir.img\n")               # backslash path
    (tmp_path / "m.scene").write_text("scene m.obj\n")
    hs = P.HostScene.load(str(tmp_path / "m.scene"), image_loader=provider)
    def tex(i):
        t = hs.textures[int(i)]
        o, w, h, c = int(t["offset"]), int(t["w"]), int(t["h"]), int(t["nb_chan"])
        return hs.texels[o: o + w * h * c].reshape(h, w, c)

    m = hs.materials
    # m0: 1x1 unit (Kd, mean Ks)
    np.testing.assert_array_equal(tex(m[0]["diffuse_spec_map"]).ravel(), np.float32([0.1, 0.2, 0.3, np.float32(np.float64(np.float32(0.3) * 3) / 3.0)]))
    # m1: rgb + default alpha
    t = tex(m[1]["diffuse_spec_map"])
    np.testing.assert_array_equal(t[..., :3], imgs["rgb4x2.img"])
    assert (t[..., 3] == np.float32(np.float64(np.float32(0.6) + np.float32(0.6) + np.float32(0.6)) / 3.0)).all()
    # m2: default rgb + alpha map
    t = tex(m[2]["diffuse_spec_map"])
    np.testing.assert_array_equal(t[..., 3], imgs["a4x2.img"][..., 0])
    np.testing.assert_array_equal(t[0, 0, :3], np.float32([0.4, 0.5, 0.6]))
    # m3/m4: packed once, normal map registered once (ids shared), normal stored as 3 channels
    assert m[3]["diffuse_spec_map"] == m[4]["diffuse_spec_map"] and m[3]["normal_map"] == m[4]["normal_map"] >= 0
    t = tex(m[3]["diffuse_spec_map"])
    np.testing.assert_array_equal(t[..., :3], imgs["rgb4x2.img"])
    np.testing.assert_array_equal(t[..., 3], imgs["a4x2.img"][..., 0])
    np.testing.assert_array_equal(tex(m[3]["normal_map"]), imgs["nrm3x3.img"])
    # m5: failed diffuse -> unit; 1-channel "normal map" dropped (the reference would read out of bounds)
    assert hs.textures[m[5]["diffuse_spec_map"]]["w"] == 1 and m[5]["normal_map"] == -1
    # m6: rgb failed, alpha present -> alpha map with default rgb
    t = tex(m[6]["diffuse_spec_map"])
    assert t.shape == (2, 4, 4) and (t[..., 0] == np.float32(0.2)).all()
    np.testing.assert_array_equal(t[..., 3], imgs["a4x2.img"][..., 0])
    # m7: alpha 2x1 brought to 4x2 with stbir_resize_float's arithmetic (material_loader.cpp:350-375; P.resize_float is held
    # to the real stb_image_resize in test_ref_thirdparty.py)
    t = tex(m[7]["diffuse_spec_map"])
    assert t.shape == (2, 4, 4)
    np.testing.assert_array_equal(t[:, :, :3], imgs["rgb4x2.img"])
    np.testing.assert_array_equal(t[:, :, 3].view(np.uint32), P.resize_float(imgs["a2x1.img"], 4, 2)[:, :, 0].view(np.uint32))
    # m8: backslash path fails on Linux unless normalised
    assert hs.textures[m[8]["diffuse_spec_map"]]["w"] == 1 and "sub\\dir.img" in hs.unloaded_textures and "missing.img" in hs.unloaded_textures
    hs2 = P.HostScene.load(str(tmp_path / "m.scene"), normalise_backslashes=True, image_loader=provider)
    o8 = int(hs2.textures[int(hs2.materials[8]["diffuse_spec_map"])]["offset"])
    np.testing.assert_array_equal(hs2.texels[o8: o8 + 3], imgs["sub/dir.img"].ravel())
    assert calls.count(str(tmp_path) + "//rgb4x2.img") == 2      # decoded once per scene load (_loaded_tex), two loads


def test_ldr_to_float_and_real_assets(P):
    """stbi_loadf's conversion rule and the shipped crate_land textures / cube cross."""
    img = np.array([[[0, 128, 255, 64]]], dtype=np.uint8)
    f = P.ldr_to_float(img)
    np.testing.assert_allclose(f[0, 0, :3], [0.0, (128 / 255) ** 2.2, 1.0], rtol=1e-6)
    assert f[0, 0, 3] == np.float32(64) / np.float32(255)          # alpha stays linear
    assert P.ldr_to_float(np.array([[7]], dtype=np.uint8)).shape == (1, 1, 1)
    hs = P.HostScene.load(os.path.join(ASSETS, "crate_land.scene"))          # built-in decoder
    assert hs.unloaded_textures == []
    assert [(t["w"], t["h"], t["nb_chan"]) for t in hs.textures] == [(1024, 1024, 4), (1024, 1024, 3)] * 2
    assert list(hs.materials["normal_map"]) == [1, 3]
    cube = P.cubemap_for_scene(hs, asset_folder=ASSETS)
    assert cube.shape == (6, 1024, 1024, 4) and (cube[..., 3] == 0).all() and 0.05 < cube[..., :3].mean() < 0.9
    # an injected provider (PIL's libjpeg) gives the same layout; its pixels differ from stb's by a few LSB at most
    hp = P.HostScene.load(os.path.join(ASSETS, "crate_land.scene"), image_loader=P.pil_image_loader)
    assert hp.textures.tolist() == hs.textures.tolist()
    assert np.abs(hp.texels - hs.texels).max() < 0.05 and (hp.texels != hs.texels).any()
    # with decoding off the same scene degrades to constants
    assert len(P.HostScene.load(os.path.join(ASSETS, "crate_land.scene"), decode_images=False).texels) == 8
    # indoor.mtl names its textures with backslashes: not found on Linux (reference behaviour) unless normalised
    ind = P.HostScene.load(os.path.join(ASSETS, "indoor.scene"), normalise_backslashes=True)
    got = {(int(t["w"]), int(t["h"]), int(t["nb_chan"])) for t in ind.textures}
    assert (512, 512, 4) in got and (512, 512, 3) in got        # wooden_planck/albedo_2.jpg, crack2.jpg
    assert (1024, 1024, 4) in got and (1024, 1024, 3) in got    # parquet / concrete albedo + normal maps
    assert ind.unloaded_textures == []
    # the reference's own behaviour on Linux: none of them is found, every material degrades to its 1x1 constant
    ref = P.HostScene.load(os.path.join(ASSETS, "indoor.scene"))
    assert "textures\\parquet\\albedo_1.jpg" in ref.unloaded_textures and all(int(t["w"]) == 1 for t in ref.textures)
