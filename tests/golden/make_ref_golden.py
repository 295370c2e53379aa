"""Regenerates tests/golden/ref_thirdparty.json FROM THE REFERENCE'S OWN THIRD-PARTY CODE.

Needs oracle/_ref/libref_thirdparty.so (`make oracle-ref`: tinyobj 1.0.8 + stb_image 2.16 compiled from where
they lie under /root/reference).  The fixture holds only outputs (digests and bit patterns):

  jpeg[rel_path]        shape and sha256 of stbi_load / stbi_loadf output for every JPEG under assets/
  ldr_to_linear_bits    stbi_loadf's float for each 8-bit level (its pow(v/255, 2.2f); level-exact gray JPEGs)
  obj[name]             sha256 of tinyobj::LoadObj's output flattened as scene.cpp:218-262 does, per shipped OBJ
  decimal_bits          float bits tinyobj's decimal reader produces for DECIMAL_SPELLINGS
  hdr_sha256            sha256 over stbi_loadf's output for the seeded Radiance files of test_ref_thirdparty.hdr_cases(seed=3, n=25)
  png_sha256            sha256 over stbi_load's output for the seeded PNGs of test_ref_thirdparty.png_cases(reps=2, seed=9)
  resize_sha256         sha256 over stbir_resize_float's outputs for the seeded cases of test_ref_thirdparty.resize_cases

so that the loader / decoder stay pinned where /root/reference is absent (tests/test_ref_thirdparty.py).
"""
import glob
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ASSETS = os.path.join(ROOT, "assets")

OBJ_ASSETS = ["indoor", "crate_land", "color_sample", "island", "sss_crate"]

# one `v <spelling> 0 0` line each; tinyobj rejects some of them (-> 0.0), truncates others at the first bad character
DECIMAL_SPELLINGS = [
    "0", "-0", "+0", "1", "-1", "+1.5", "0.1", "0.2", "0.3", "0.7", "1.1", "2.675", "3.14159265358979", "0.000001", "0.0000001",
    "0.00000001", "0.123456789", "0.1234567890123456789", "123456789.123456789", "1e3", "1E3", "1e-3", "1e+3", "-2.5e+2", "6.02e23",
    "1.6e-19", "1e38", "1e39", "1e-45", "1e-46", "4.9e-324", "1.7976931348623157e308", ".5", "-.5", "5.", "5.e2", "1e", "1e+", "1ex",
    "1.5abc", "abc", "1..2", "1.2.3", "--1", "+-1", "0x10", "inf", "nan", "1,5", "1_000", "00012", "0.30000000000000004",
    "16777217", "16777219", "0.1e1", "100e-2", "9007199254740993", "1e22", "1e23", "123456789012345678901234567890", "3.4028235e38",
    "3.4028236e38", "1.17549435e-38", "1e-40", "7.0064923216240854e-46", "0.999999970197677612", "1.00000011920928955",
]


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def jpeg_assets():
    return sorted(glob.glob(os.path.join(ASSETS, "**", "*.jpg"), recursive=True))


def decimal_obj_text() -> str:
    lines = ["mtllib f.mtl"] + ["v %s 0 0" % s for s in DECIMAL_SPELLINGS] + ["usemtl a"]
    n = len(DECIMAL_SPELLINGS)
    lines += ["f %d %d %d" % (i + 1, i + 1, i + 1) for i in range(n)]       # face i's first vertex = spelling i
    return "\n".join(lines) + "\n"


def _digest(mesh_sizes, vertices, normals, texcoords, material_ids, iors, kds, names) -> str:
    h = hashlib.sha256()
    h.update(np.asarray(mesh_sizes, np.uint32).tobytes())
    for a in (vertices, normals, texcoords):
        h.update(np.ascontiguousarray(a, np.float32).tobytes())
    h.update(np.asarray(material_ids, np.int32).tobytes())
    h.update(np.asarray(iors, np.float32).tobytes())
    h.update(np.ascontiguousarray(kds, np.float32).tobytes())
    h.update("\n".join(names).encode())
    return h.hexdigest()


def my_flatten_digest(hs) -> str:
    """The same digest from a HostScene loaded with decode_images=False (every texture is then its 1x1 Kd/Ks constant)."""
    kds = np.array([hs.texels[int(hs.textures[int(m["diffuse_spec_map"])]["offset"]):][:3] for m in hs.materials], np.float32)
    return _digest(hs.mesh_sizes, hs.faces["vertices"], hs.faces["normals"], hs.faces["texcoords"], hs.faces["material_id"],
                   hs.materials["ior"], kds, [])


if __name__ == "__main__":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import ref_thirdparty as R
    if not R.available():
        sys.exit("oracle/_ref/libref_thirdparty.so is missing: run `make oracle-ref` where /root/reference exists")
    out = {"_generated_by": "tests/golden/make_ref_golden.py from tinyobj 1.0.8 + stb_image 2.16 under /root/reference",
           "jpeg": {}, "obj": {}}
    for path in jpeg_assets():
        a8, af = R.stbi_load(path), R.stbi_loadf(path)
        out["jpeg"][os.path.relpath(path, ASSETS)] = {"shape": list(a8.shape), "sha256_u8": sha(a8), "sha256_f32": sha(af)}
    from PIL import Image
    table = np.zeros(256, np.uint32)
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "flat.jpg")
        for v in range(256):
            Image.fromarray(np.full((8, 8), v, np.uint8)).save(p, "JPEG", quality=100)
            a8, af = R.stbi_load(p), R.stbi_loadf(p)
            table[int(a8[0, 0, 0])] = af[0, 0, 0].view(np.uint32)
        out["ldr_to_linear_bits"] = [int(x) for x in table]
        for name in OBJ_ASSETS:
            d, _ = R.tinyobj_load(os.path.join(ASSETS, "obj", name + ".obj"), os.path.join(ASSETS, "obj") + "/")
            meshes = R.flatten_like_reference(d)
            # names are left out of the committed digest (HostScene does not carry them); the live test checks them
            out["obj"][name] = _digest([len(m["material_ids"]) for m in meshes],
                                       np.concatenate([np.nan_to_num(m["vertices"]) for m in meshes]),
                                       np.concatenate([np.nan_to_num(m["normals"]) for m in meshes]),
                                       np.concatenate([np.nan_to_num(m["texcoords"]) for m in meshes]),
                                       np.concatenate([m["material_ids"] for m in meshes]),
                                       [m["ior"] for m in d["materials"]],
                                       np.array([m["diffuse"] for m in d["materials"]], np.float32), [])
        with open(os.path.join(tmp, "f.obj"), "w") as f:
            f.write(decimal_obj_text())
        with open(os.path.join(tmp, "f.mtl"), "w") as f:
            f.write("newmtl a\n")
        d, msg = R.tinyobj_load(os.path.join(tmp, "f.obj"), tmp + "/")
        assert d is not None, msg
        xs = d["vertices"].reshape(-1, 3)[:, 0]
        assert len(xs) == len(DECIMAL_SPELLINGS)
        out["decimal_bits"] = [int(b) for b in xs.view(np.uint32)]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_ref_thirdparty import resize_cases, png_cases, hdr_cases
    h = hashlib.sha256()
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "t.hdr")
        for name, data in hdr_cases(seed=3, n=25):
            with open(p, "wb") as f:
                f.write(data)
            img = R.stbi_loadf(p)
            assert img is not None, name
            h.update(np.array(img.shape, np.int32).tobytes() + img.tobytes())
    out["hdr_sha256"] = h.hexdigest()
    h = hashlib.sha256()
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "t.png")
        for name, data in png_cases(reps=2, seed=9):
            with open(p, "wb") as f:
                f.write(data)
            img = R.stbi_load(p)
            assert img is not None, name
            h.update(np.array(img.shape, np.int32).tobytes() + img.tobytes())
    out["png_sha256"] = h.hexdigest()
    h = hashlib.sha256()
    for img, ow, oh in resize_cases():
        h.update(R.stbir_resize_float(img, ow, oh).tobytes())
    out["resize_sha256"] = h.hexdigest()
    with open(os.path.join(HERE, "ref_thirdparty.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote ref_thirdparty.json:", len(out["jpeg"]), "jpegs,", len(out["obj"]), "objs,", len(out["decimal_bits"]), "decimals;",
          "levels seen in the table:", int((table != 0).sum()))
