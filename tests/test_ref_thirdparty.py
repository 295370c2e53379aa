"""The input side of the hot path against THE REFERENCE'S OWN third-party code.

oracle/_ref/libref_thirdparty.so is tinyobj 1.0.8 + stb_image 2.16 compiled from where they lie under
/root/reference (`make oracle-ref`); this is the one part of the reference that builds here without
stand-ins, so these are the only parity tests in the repository pinned by reference-built code:

  * host/scene_loader.cpp's OBJ/MTL parse  == tinyobj::LoadObj (+ the flatten of scene.cpp:218-262), bit for bit,
    on the shipped assets and on fuzzed OBJ/MTL text (float spellings, negative indices, polygons, groups,
    usemtl switches, texture options, CRLF / lone-CR line ends);
  * host/image_decode.cpp's JPEG decode    == stbi_load / stbi_loadf, bit for bit, on the shipped JPEGs and on
    synthetic ones covering the sampling layouts, progressive/baseline, restart intervals, odd sizes, CMYK.

Where the library is absent (no /root/reference) the live tests skip and the committed fixture
tests/golden/ref_thirdparty.json (generated from it by tests/golden/make_ref_golden.py) is what pins the
same code paths.
"""
import hashlib
import io
import json
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import ASSETS, GOLDEN
from golden.make_ref_golden import DECIMAL_SPELLINGS, OBJ_ASSETS, jpeg_assets, sha, my_flatten_digest, decimal_obj_text

import ref_thirdparty as R

live = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")


# ------------------------------------------------------------------ OBJ / MTL

def load_mine(P, obj_path):
    """Loads obj_path through a temporary one-line .scene next to it (textures not decoded)."""
    sc = os.path.join(os.path.dirname(obj_path), "_cmp_%d.scene" % os.getpid())
    with open(sc, "w") as f:
        f.write("scene %s\n" % os.path.basename(obj_path))
    try:
        return P.HostScene.load(sc, decode_images=False)
    finally:
        os.remove(sc)


def compare_with_tinyobj(P, obj_path, mtl_dir):
    d, msg = R.tinyobj_load(obj_path, mtl_dir)
    try:
        hs, mine_err = load_mine(P, obj_path), None
    except P.native.PtamdError as e:
        hs, mine_err = None, str(e)
    if d is None:
        assert hs is None, "tinyobj rejects the file (%s) but the loader accepted it" % msg.strip()
        return "both reject"
    meshes = R.flatten_like_reference(d)
    if hs is None:
        # the one deliberate difference: a face whose material id is -1 makes the reference index
        # materials[-1]; the loader refuses the scene instead
        assert "valid material" in mine_err and any((m["material_ids"] < 0).any() for m in meshes), mine_err
        return "refused (-1 material)"
    assert [len(m["material_ids"]) for m in meshes] == list(hs.mesh_sizes)
    off = 0
    for m in meshes:
        f = hs.faces[off: off + len(m["material_ids"])]
        off += len(f)
        np.testing.assert_array_equal(f["vertices"].view(np.uint32), m["vertices"].view(np.uint32))
        for key in ("normals", "texcoords"):        # corners without an index: reference reads out of bounds, loader zeros
            ok = ~np.isnan(m[key])
            np.testing.assert_array_equal(f[key][ok].view(np.uint32), m[key][ok].view(np.uint32))
            assert (f[key][~ok] == 0).all()
        np.testing.assert_array_equal(f["material_id"].astype(np.int64), m["material_ids"])
    assert len(d["materials"]) == len(hs.materials)
    expected_names = []
    for i, m in enumerate(d["materials"]):
        assert np.float32(m["ior"]).view(np.uint32) == hs.materials[i]["ior"].view(np.uint32)
        t = hs.textures[hs.materials[i]["diffuse_spec_map"]]
        rgba = hs.texels[int(t["offset"]): int(t["offset"]) + 4]            # decode off: the 1x1 (Kd, mean Ks) constant
        np.testing.assert_array_equal(rgba[:3].view(np.uint32), m["diffuse"].view(np.uint32))
        ks = np.float32((np.float64(m["specular"][0] + m["specular"][1] + m["specular"][2])) / 3.0)
        assert rgba[3].view(np.uint32) == ks.view(np.uint32)
        for name in (m["diffuse_texname"], m["specular_texname"], m["bump_texname"] or m["normal_texname"]):
            if name and name not in expected_names:
                expected_names.append(name)
    assert sorted(hs.unloaded_textures) == sorted(expected_names)           # the texture NAMES the MTL parse produced
    return "same"


@live
@pytest.mark.parametrize("name", OBJ_ASSETS)
def test_loader_matches_tinyobj_on_shipped_assets(P, name):
    assert compare_with_tinyobj(P, os.path.join(ASSETS, "obj", name + ".obj"), os.path.join(ASSETS, "obj") + "/") == "same"


@live
def test_loader_matches_tinyobj_on_the_generated_atrium(P, tmp_path):
    """BASELINE.json configs[3] asset (cuda_pathtracer_amd.synthetic.write_atrium: 264 832 triangles in 298 objects,
    27 MB of OBJ text): the loader's parse + flatten == the reference's tinyobj + scene.cpp:218-262, bit for bit."""
    from cuda_pathtracer_amd.synthetic import write_atrium
    scene = write_atrium(str(tmp_path))
    obj = os.path.join(os.path.dirname(scene), "obj", "atrium.obj")
    assert compare_with_tinyobj(P, obj, os.path.dirname(obj) + "/") not in ("both reject", "refused (-1 material)")
    hs = P.HostScene.load(scene)
    assert len(hs.faces) == 264832 and len(hs.mesh_sizes) == 298 and len(hs.lights) == 6
    # deterministic: the same seed writes the same bytes
    again = write_atrium(str(tmp_path / "again"))
    with open(obj, "rb") as f1, open(os.path.join(os.path.dirname(again), "obj", "atrium.obj"), "rb") as f2:
        assert hashlib.sha256(f1.read()).digest() == hashlib.sha256(f2.read()).digest()


def _spell(rng, x):
    k = int(rng.integers(0, 9))
    if k == 6:
        s = "%.5f" % x
        return s.replace("0.", ".", 1) if s.startswith(("0.", "-0.")) else s        # ".5": tinyobj rejects -> default
    return [repr(float(x)), "%.3f" % x, "%e" % x, "%.10g" % x, "%+.4f" % x, "%.2E" % x, None, str(int(round(x))),
            "%.17g" % x][k]


def fuzz_obj(rng, eol):
    nv, nn, nt = int(rng.integers(3, 30)), int(rng.integers(1, 10)), int(rng.integers(1, 10))
    ws = lambda: str(rng.choice([" ", "  ", "\t", " \t "]))
    L = ["# fuzz", "mtllib f.mtl", ""]
    for _ in range(nv):
        c = rng.normal(size=3) * 10.0 ** int(rng.integers(-3, 3))
        extra = (ws() + _spell(rng, 1.0)) if rng.random() < 0.1 else ""
        L.append(str(rng.choice(["", " ", "\t"])) + "v" + ws() + ws().join(_spell(rng, x) for x in c) + extra
                 + str(rng.choice(["", " ", "  # c"])))
    for _ in range(nn):
        L.append("vn" + ws() + ws().join(_spell(rng, x) for x in rng.normal(size=3)))
    for _ in range(nt):
        L.append("vt" + ws() + ws().join(_spell(rng, x) for x in rng.uniform(-2, 3, size=int(rng.choice([1, 2, 2, 2, 3])))))
    mats = ["m%d" % i for i in range(int(rng.integers(1, 4)))]
    L.append("usemtl " + mats[0])
    for f in range(int(rng.integers(1, 25))):
        r = rng.random()
        if r < 0.12:
            L.append(str(rng.choice(["g", "o"])) + " grp%d" % f + str(rng.choice(["", " other"])))
        if r > 0.85:
            L.append("usemtl " + str(rng.choice(mats + (["nope"] if rng.random() < 0.1 else []))))
        if rng.random() < 0.05:
            L.append("s " + str(rng.choice(["off", "1"])))
        style = int(rng.integers(0, 4))

        def ix(n):
            i = int(rng.integers(1, n + 1))
            return str(i) if rng.random() < 0.8 else str(i - n - 1)
        corners = []
        for _ in range(int(rng.choice([3, 3, 3, 4, 5, 6]))):
            v = ix(nv)
            corners.append([v, v + "/" + ix(nt), v + "//" + ix(nn), v + "/" + ix(nt) + "/" + ix(nn)][style])
        L.append("f" + ws() + ws().join(corners) + str(rng.choice(["", " "])))
    M = []
    for m in mats:
        M.append("newmtl " + m)
        if rng.random() < 0.9: M.append("Kd " + " ".join(_spell(rng, x) for x in rng.uniform(0, 1, 3)))
        if rng.random() < 0.7: M.append("Ks " + " ".join(_spell(rng, x) for x in rng.uniform(0, 1, 3)))
        if rng.random() < 0.5: M.append("Ni " + _spell(rng, rng.uniform(1, 2)))
        if rng.random() < 0.3: M.append("d " + _spell(rng, rng.uniform(0, 1)))
        if rng.random() < 0.3: M.append("illum %d" % rng.integers(0, 8))
        if rng.random() < 0.4:
            M.append("map_Kd " + str(rng.choice(["", "-bm 0.5 ", "-clamp on ", "-s 1 2 3 ", "-s 1 2 ", "-o 1 ", "-mm 0 1 ", "-type sphere ",
                                                 "-imfchan r ", "-unknown 3 ", "-blendu off -boost 2 "]))
                     + str(rng.choice(["tex.jpg", "a b.jpg", "dir/t.jpg", "dir\\t.jpg"])))
        if rng.random() < 0.4:
            M.append(str(rng.choice(["map_Bump", "map_bump", "bump", "norm"])) + " " + str(rng.choice(["", "-bm 2 "])) + "n.jpg")
        if rng.random() < 0.2: M.append("map_Ks s.jpg" + str(rng.choice(["", "  ", "\t"])))
        if rng.random() < 0.2: M.append("weird_param 1 2 3")
        M.append("")
    return eol.join(L) + eol, eol.join(M) + eol


@live
def test_loader_matches_tinyobj_on_fuzzed_text(P, tmp_path):
    outcomes = {}
    for seed in range(160):
        rng = np.random.default_rng(seed)
        obj, mtl = fuzz_obj(rng, ["\n", "\n", "\n", "\r\n", "\r"][seed % 5])
        d = tmp_path / ("c%d" % seed)
        d.mkdir()
        (d / "f.obj").write_bytes(obj.encode())
        (d / "f.mtl").write_bytes(mtl.encode())
        r = compare_with_tinyobj(P, str(d / "f.obj"), str(d) + "/")
        outcomes[r] = outcomes.get(r, 0) + 1
    assert outcomes.get("same", 0) >= 120, outcomes


@live
def test_loader_mtl_corner_cases_match_tinyobj(P, tmp_path):
    cases = {
        "no_newmtl": ("mtllib f.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl \nf 1 2 3\n", "Kd 0.25 0.5 0.75\n"),       # default material ""
        "empty_mtl": ("mtllib f.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl \nf 1 2 3\n", ""),
        "two_blanks_in_mtllib": ("mtllib  f.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl \nf 1 2 3\n", "newmtl a\nKd 1 1 1\n"),
        "duplicate_names": ("mtllib f.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n", "newmtl a\nKd 1 0 0\nnewmtl a\nKd 0 1 0\n"),
        "second_mtllib_file": ("mtllib missing.mtl f.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n", "newmtl a\nNi 1.5\n"),
        "name_with_blanks": ("mtllib f.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl  a b\nf 1 2 3\n", "newmtl  a b  \nKd 1 1 1\n"),
        "o_drops_usemtl_exported_faces": ("mtllib f.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\nusemtl b\no x\nf 3 2 1\n",
                                          "newmtl a\nnewmtl b\n"),
        "g_keeps_them": ("mtllib f.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\nusemtl b\ng x\nf 3 2 1\n", "newmtl a\nnewmtl b\n"),
        "zero_index": ("mtllib f.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 0 1 2\n", "newmtl a\n"),
        "exponents": ("mtllib f.mtl\nv 1e3 1E-3 -2.5e+2\nv 1e 5 6\nv 7.e2 8.5e 9e-\nusemtl a\nf 1 2 3\n", "newmtl a\nKd 1e-1 .5 5.\n"),
    }
    seen = set()
    for name, (obj, mtl) in cases.items():
        d = tmp_path / name
        d.mkdir()
        (d / "f.obj").write_text(obj)
        (d / "f.mtl").write_text(mtl)
        seen.add(compare_with_tinyobj(P, str(d / "f.obj"), str(d) + "/"))
    assert seen == {"same", "both reject"}


# ------------------------------------------------------------------ JPEG

def synthetic_jpegs():
    """(name, bytes) pairs made with PIL's encoder: every sampling layout stb special-cases, both entropy modes,
    restart intervals, optimised tables, tiny and ragged sizes, grayscale, CMYK, RGB-identified components."""
    from PIL import Image
    rng = np.random.default_rng(7)
    out = []

    def picture(w, h, c):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 5 + yy * 3) % 256, (xx * yy) % 256, (255 - xx * 2 - yy) % 256, (xx + yy * 7) % 256], axis=2)[:, :, :c]
        noise = rng.integers(-40, 40, size=(h, w, c))
        return np.clip(base + noise, 0, 255).astype(np.uint8)

    def enc(name, arr, mode, **kw):
        buf = io.BytesIO()
        Image.frombytes(mode, (arr.shape[1], arr.shape[0]), np.ascontiguousarray(arr).tobytes()).save(buf, "JPEG", **kw)
        out.append((name, buf.getvalue()))

    for (w, h) in ((1, 1), (7, 5), (16, 16), (17, 33), (100, 57), (64, 8)):
        for sub in (0, 1, 2):
            for prog in (False, True):
                enc("rgb_%dx%d_s%d_%s" % (w, h, sub, "p" if prog else "b"), picture(w, h, 3), "RGB", quality=int(rng.integers(30, 96)),
                    subsampling=sub, progressive=prog, optimize=bool(rng.integers(0, 2)))
        enc("gray_%dx%d" % (w, h), picture(w, h, 1), "L", quality=80)
        enc("gray_%dx%d_p" % (w, h), picture(w, h, 1), "L", quality=60, progressive=True)
    enc("q100", picture(40, 40, 3), "RGB", quality=100, subsampling=0)
    enc("q1", picture(40, 40, 3), "RGB", quality=1, subsampling=2)
    enc("restart_blocks", picture(90, 70, 3), "RGB", quality=75, subsampling=2, restart_marker_blocks=3)
    enc("restart_rows_prog", picture(90, 70, 3), "RGB", quality=75, subsampling=1, restart_marker_rows=1, progressive=True)
    enc("restart_gray", picture(50, 50, 1), "L", quality=75, restart_marker_blocks=1)
    enc("cmyk", picture(33, 21, 4), "CMYK", quality=85)
    enc("cmyk_prog", picture(33, 21, 4), "CMYK", quality=85, progressive=True)
    enc("keep_rgb", picture(33, 21, 3), "RGB", quality=85, keep_rgb=True)
    return out


@live
def test_jpeg_decoder_matches_stb_on_shipped_files(P):
    files = jpeg_assets()
    assert len(files) >= 9
    for path in files:
        ref8 = R.stbi_load(path)
        np.testing.assert_array_equal(P.load_image8(path), ref8, err_msg=path)
        np.testing.assert_array_equal(P.load_image(path).view(np.uint32), R.stbi_loadf(path).view(np.uint32), err_msg=path)


@live
def test_jpeg_decoder_matches_stb_on_synthetic_files(P, tmp_path):
    n = 0
    for name, data in synthetic_jpegs():
        path = str(tmp_path / (name + ".jpg"))
        with open(path, "wb") as f:
            f.write(data)
        ref8 = R.stbi_load(path)
        assert ref8 is not None, name
        np.testing.assert_array_equal(P.load_image8(path), ref8, err_msg=name)
        n += 1
    assert n > 50


@live
def test_reference_libm_float_conversion_table(P, tmp_path):
    """stbi_loadf's pow(v / 255, 2.2f) for all 256 byte values: the loader's table == stb's output."""
    from PIL import Image
    path = str(tmp_path / "ramp.jpg")
    Image.fromarray(np.arange(256, dtype=np.uint8).reshape(16, 16).repeat(8, 0).repeat(8, 1)).save(path, "JPEG", quality=100)
    a8, af = R.stbi_load(path), R.stbi_loadf(path)
    m8, mf = P.load_image8(path), P.load_image(path)
    np.testing.assert_array_equal(m8, a8)
    np.testing.assert_array_equal(mf.view(np.uint32), af.view(np.uint32))
    assert len(np.unique(a8)) > 200


# ------------------------------------------------------------------ PNG

def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def _pack_rows(samples, depth):
    rows = []
    for s in samples:
        if depth == 8:
            rows.append(bytes(s.astype(np.uint8)))
        elif depth == 16:
            rows.append(s.astype(">u2").tobytes())
        else:
            per = 8 // depth
            s2 = np.concatenate([s, np.zeros((-len(s)) % per, s.dtype)]).reshape(-1, per)
            v = np.zeros(len(s2), np.int64)
            for k in range(per):
                v = (v << depth) | s2[:, k]
            rows.append(bytes(v.astype(np.uint8)))
    return rows


def _filter_rows(rows, bpp, rng):
    """Applies a random PNG filter type (0..4) to every row."""
    out, prev = bytearray(), None
    for r in rows:
        ft = int(rng.integers(0, 5))
        cur = np.frombuffer(r, np.uint8).astype(np.int64)
        zero = np.zeros(len(cur), np.int64)
        shift = lambda v: np.concatenate([np.zeros(bpp, np.int64), v[:-bpp]]) if len(v) > bpp else zero
        a, b = shift(cur), (prev if prev is not None else zero)
        c = shift(b)
        if ft == 0: f = cur
        elif ft == 1: f = cur - a
        elif ft == 2: f = cur - b
        elif ft == 3: f = cur - ((a + b) >> 1)
        else:
            p = a + b - c
            pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
            f = cur - np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        out.append(ft)
        out += bytes((f & 255).astype(np.uint8))
        prev = cur
    return bytes(out)


def make_png(rng, w, h, colour, depth, interlace, trns):
    """A PNG of the given colour type / bit depth / interlace mode with random pixels, random filter per row, optional
    tRNS (palette alpha, or a colour key that really occurs), IDAT split in two, ancillary chunks sprinkled in."""
    n = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[colour]
    pal = None
    if colour == 3:
        pal = rng.integers(0, 256, size=(int(rng.integers(1, min(256, 1 << depth) + 1)), 3)).astype(np.uint8)
        img = rng.integers(0, len(pal), size=(h, w, 1))
    else:
        img = rng.integers(0, 1 << depth, size=(h, w, n))
        if rng.random() < 0.5:
            img[: h // 2 + 1] = img[0, 0]
    bpp = 1 if depth < 8 else n * depth // 8
    if not interlace:
        data = _filter_rows(_pack_rows(img.reshape(h, w * n), depth), bpp, rng)
    else:
        data = b""
        for x0, y0, dx, dy in zip((0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)):
            sub = img[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                data += _filter_rows(_pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, rng)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, interlace))
    if rng.random() < 0.3:
        out += _chunk(b"gAMA", struct.pack(">I", 45455))
    if pal is not None:
        out += _chunk(b"PLTE", pal.tobytes())
        if trns:
            out += _chunk(b"tRNS", bytes(rng.integers(0, 256, size=int(rng.integers(1, len(pal) + 1))).astype(np.uint8)))
    elif trns and colour in (0, 2):
        out += _chunk(b"tRNS", b"".join(struct.pack(">H", int(v)) for v in img[0, 0]))
    z = zlib.compress(data, int(rng.choice([0, 1, 6, 9])))
    cut = int(rng.integers(1, len(z))) if len(z) > 1 and rng.random() < 0.5 else len(z)
    out += _chunk(b"IDAT", z[:cut])
    if cut < len(z):
        out += _chunk(b"IDAT", z[cut:])
    if rng.random() < 0.3:
        out += _chunk(b"tEXt", b"k\0v")
    return out + _chunk(b"IEND", b"")


PNG_KINDS = [(0, d) for d in (1, 2, 4, 8, 16)] + [(2, 8), (2, 16)] + [(3, d) for d in (1, 2, 4, 8)] + [(4, 8), (4, 16), (6, 8), (6, 16)]


def png_cases(reps=3, seed=0):
    rng = np.random.default_rng(seed)
    for rep in range(reps):
        for colour, depth in PNG_KINDS:
            for interlace in (0, 1):
                for trns in (False, True):
                    w, h = (1, 1) if rep == 0 else (int(v) for v in rng.integers(1, 40, size=2))
                    yield "c%dd%di%dt%d_%dx%d" % (colour, depth, interlace, trns, w, h), make_png(rng, w, h, colour, depth, interlace, trns)


@live
def test_png_decoder_matches_stb(P, tmp_path):
    path = str(tmp_path / "t.png")
    n = 0
    for name, data in png_cases():
        with open(path, "wb") as f:
            f.write(data)
        ref8 = R.stbi_load(path)
        assert ref8 is not None, name
        np.testing.assert_array_equal(P.load_image8(path), ref8, err_msg=name)
        if n % 7 == 0:          # float rule incl. the linear alpha channel of 2- and 4-channel images
            np.testing.assert_array_equal(P.load_image(path).view(np.uint32), R.stbi_loadf(path).view(np.uint32), err_msg=name)
        n += 1
    assert n == 3 * len(PNG_KINDS) * 4
    from PIL import Image                                   # and an ordinary encoder's output
    rng = np.random.default_rng(4)
    for mode, c in (("L", 1), ("LA", 2), ("RGB", 3), ("RGBA", 4), ("P", 1)):
        arr = rng.integers(0, 256, size=(37, 53, c), dtype=np.uint8)
        Image.frombytes(mode, (53, 37), arr.tobytes()).save(path, "PNG", optimize=bool(c & 1))
        np.testing.assert_array_equal(P.load_image8(path), R.stbi_load(path), err_msg=mode)


def test_fixture_png_digest(P, fixture, tmp_path):
    path = str(tmp_path / "t.png")
    h = hashlib.sha256()
    for name, data in png_cases(reps=2, seed=9):
        with open(path, "wb") as f:
            f.write(data)
        img = P.load_image8(path)
        h.update(np.array(img.shape, np.int32).tobytes() + img.tobytes())
    assert h.hexdigest() == fixture["png_sha256"]


def test_png_decoder_survives_corrupt_input(P, tmp_path):
    rng = np.random.default_rng(8)
    path = str(tmp_path / "c.png")
    ok = errors = 0
    for name, data in png_cases(reps=2, seed=5):
        d = bytearray(data)
        k = int(rng.integers(0, 3))
        if k == 0:
            d = d[: int(rng.integers(8, len(d)))]
        else:
            for _ in range(int(rng.integers(1, 6))):
                d[int(rng.integers(8, len(d)))] = int(rng.integers(0, 256))
        with open(path, "wb") as f:
            f.write(d)
        try:
            assert P.load_image8(path).size > 0
            ok += 1
        except P.native.PtamdError:
            errors += 1
    assert ok + errors == 2 * len(PNG_KINDS) * 4 and errors > 20


# ------------------------------------------------------------------ Radiance HDR (stbi_loadf's float path)

def _rle_channel(vals, rng):
    out, i, n = bytearray(), 0, len(vals)
    while i < n:
        j = i
        while j < n and vals[j] == vals[i] and j - i < 127:
            j += 1
        if j - i >= 3 or rng.random() < 0.2:
            out += bytes([128 + (j - i), int(vals[i])])
            i = j
        else:
            k = min(n, i + int(rng.integers(1, 128)))
            out += bytes([k - i]) + bytes(int(v) for v in vals[i:k])
            i = k
    return bytes(out)


def make_hdr(rng, w, h, rle, ident, extra):
    px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    px[rng.random((h, w)) < 0.2, 3] = 0                         # zero exponent: black
    if rng.random() < 0.5:
        px[:, : w // 2] = px[0, 0]                              # runs
    head = ident + b"\n" + (b"# comment\nEXPOSURE=1.0\n" if extra else b"") + b"FORMAT=32-bit_rle_rgbe\n\n" + ("-Y %d +X %d\n" % (h, w)).encode()
    body = bytearray()
    if rle and 8 <= w < 32768:
        for y in range(h):
            body += bytes([2, 2, (w >> 8) & 0xFF, w & 0xFF])
            for k in range(4):
                body += _rle_channel(px[y, :, k], rng)
    else:
        body += px.tobytes()                                    # flat pixels (also what stb falls back to mid-file)
    return head + bytes(body)


def hdr_cases(seed=0, n=40):
    rng = np.random.default_rng(seed)
    for i in range(n):
        w, h = int(rng.choice([1, 3, 7, 8, 9, 16, 33, 100, 257])), int(rng.integers(1, 20))
        yield "hdr%d_%dx%d" % (i, w, h), make_hdr(rng, w, h, bool(rng.integers(0, 2)), [b"#?RADIANCE", b"#?RGBE"][int(rng.integers(0, 2))],
                                               bool(rng.integers(0, 2)))


@live
def test_hdr_decoder_matches_stb(P, tmp_path):
    path = str(tmp_path / "t.hdr")
    for name, data in hdr_cases():
        with open(path, "wb") as f:
            f.write(data)
        ref = R.stbi_loadf(path)
        assert ref is not None and ref.shape[2] == 3, name
        np.testing.assert_array_equal(P.load_image(path).view(np.uint32), ref.view(np.uint32), err_msg=name)


def test_fixture_hdr_digest_and_errors(P, fixture, tmp_path):
    path = str(tmp_path / "t.hdr")
    h = hashlib.sha256()
    for name, data in hdr_cases(seed=3, n=25):
        with open(path, "wb") as f:
            f.write(data)
        img = P.load_image(path)
        h.update(np.array(img.shape, np.int32).tobytes() + img.tobytes())
    assert h.hexdigest() == fixture["hdr_sha256"]
    for junk in (b"#?RADIANCE\n", b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 2 +X 2\n" + b"\0" * 16, b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 2 +X 2\n" + b"\0" * 16,
                 b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 9\n\x02\x02\x00\x09\x85"):
        with open(path, "wb") as f:
            f.write(junk)
        with pytest.raises(P.native.PtamdError):
            P.load_image(path)


# ------------------------------------------------------------------ stbir_resize_float

def resize_cases():
    """Seeded (image, out_w, out_h) triples: growing, shrinking, mixed and equal axes, single rows/columns, 1 and 3 channels."""
    rng = np.random.default_rng(11)
    fixed = [(4, 4, 8, 8), (8, 8, 4, 4), (5, 7, 11, 13), (16, 9, 7, 20), (10, 10, 10, 20), (10, 10, 20, 10), (1, 1, 5, 5), (5, 5, 1, 1),
             (2, 9, 9, 2), (32, 32, 33, 31), (64, 16, 16, 64), (13, 1, 40, 1), (1, 13, 1, 40), (9, 9, 10, 10), (10, 10, 9, 9), (3, 200, 200, 3)]
    rnd = [tuple(int(v) for v in rng.integers(1, 70, size=4)) for _ in range(60)]
    for i, (iw, ih, ow, oh) in enumerate(fixed + rnd):
        if (iw, ih) == (ow, oh):
            continue
        c = 1 if i % 2 else 3
        img = rng.uniform(-0.5, 1.5, size=(ih, iw, c)).astype(np.float32) * np.float32(10.0 ** int(rng.integers(-3, 3)))
        yield img, ow, oh


@live
def test_resize_matches_stbir(P):
    n = 0
    for img, ow, oh in resize_cases():
        np.testing.assert_array_equal(P.resize_float(img, ow, oh).view(np.uint32), R.stbir_resize_float(img, ow, oh).view(np.uint32),
                                      err_msg="%s -> %dx%d" % (img.shape, ow, oh))
        n += 1
    assert n > 70
    big = np.random.default_rng(1).uniform(0, 1, size=(256, 128, 1)).astype(np.float32)       # a specular map grown to its albedo's size
    np.testing.assert_array_equal(P.resize_float(big, 512, 512).view(np.uint32), R.stbir_resize_float(big, 512, 512).view(np.uint32))


def test_fixture_resize_digest(P, fixture):
    h = hashlib.sha256()
    for img, ow, oh in resize_cases():
        h.update(P.resize_float(img, ow, oh).tobytes())
    assert h.hexdigest() == fixture["resize_sha256"]


# ------------------------------------------------------------------ without the reference: committed fixture + robustness

@pytest.fixture(scope="module")
def fixture():
    with open(os.path.join(GOLDEN, "ref_thirdparty.json")) as f:
        return json.load(f)


def test_fixture_jpeg_digests(P, fixture):
    """sha256 of stb_image's 8-bit and float output for every shipped JPEG (generated with oracle/_ref)."""
    for path in jpeg_assets():
        rel = os.path.relpath(path, ASSETS)
        want = fixture["jpeg"][rel]
        img8 = P.load_image8(path)
        assert list(img8.shape) == want["shape"], rel
        assert sha(img8) == want["sha256_u8"], rel
        assert sha(P.load_image(path)) == want["sha256_f32"], rel


def test_fixture_ldr_table(P, fixture, tmp_path):
    want = np.array(fixture["ldr_to_linear_bits"], dtype=np.uint32)
    from PIL import Image
    path = str(tmp_path / "flat.jpg")
    got = np.zeros(256, np.uint32)
    # a flat gray JPEG decodes to exactly its level (DC only): read the table through the public API
    for v in range(256):
        Image.fromarray(np.full((8, 8), v, np.uint8)).save(path, "JPEG", quality=100)
        img8, imgf = P.load_image8(path), P.load_image(path)
        got[int(img8[0, 0, 0])] = imgf[0, 0, 0].view(np.uint32)
    seen = got != 0
    assert seen.sum() >= 250
    np.testing.assert_array_equal(got[seen], want[seen])


def test_fixture_obj_digests(P, fixture):
    for name in OBJ_ASSETS:
        hs = load_mine(P, os.path.join(ASSETS, "obj", name + ".obj"))
        assert my_flatten_digest(hs) == fixture["obj"][name], name


def test_fixture_decimal_spellings(P, fixture, tmp_path):
    """tinyobj's decimal reader on awkward spellings: bits recorded from the real tinyobj."""
    (tmp_path / "f.obj").write_text(decimal_obj_text())
    (tmp_path / "f.mtl").write_text("newmtl a\n")
    hs = load_mine(P, str(tmp_path / "f.obj"))
    got = np.ascontiguousarray(hs.faces["vertices"][:, 0, 0]).view(np.uint32)      # x of each face's first corner
    want = np.array(fixture["decimal_bits"], dtype=np.uint32)
    assert len(want) == len(DECIMAL_SPELLINGS)
    np.testing.assert_array_equal(got, want)


def test_jpeg_decoder_survives_corrupt_input(P, tmp_path):
    """Truncations and byte flips of real files must end in a decoded picture or PtamdError, never a crash."""
    rng = np.random.default_rng(3)
    src = [os.path.join(ASSETS, "obj", "textures", "water", "normal.jpg"), os.path.join(ASSETS, "obj", "textures", "crack2.jpg")]
    path = str(tmp_path / "c.jpg")
    errors = ok = 0
    for s in src:
        data = bytearray(open(s, "rb").read())
        for trial in range(60):
            d = bytearray(data)
            if trial % 3 == 0:
                d = d[: int(rng.integers(2, len(d)))]
            else:
                for _ in range(int(rng.integers(1, 8))):
                    d[int(rng.integers(160 if trial % 3 == 1 else 2, len(d)))] = int(rng.integers(0, 256))
            with open(path, "wb") as f:
                f.write(d)
            try:
                img = P.load_image8(path)
                assert img.shape[0] > 0
                ok += 1
            except P.native.PtamdError:
                errors += 1
    assert ok + errors == 120 and errors > 5
    for junk in (b"", b"\xff", b"\xff\xd8", b"\xff\xd8\xff", b"not a jpeg at all", b"\x89PNG\r\n\x1a\n" + b"\0" * 64):
        with open(path, "wb") as f:
            f.write(junk)
        with pytest.raises(P.native.PtamdError):
            P.load_image8(path)
    with pytest.raises(P.native.PtamdError):
        P.load_image8(str(tmp_path / "missing.jpg"))
