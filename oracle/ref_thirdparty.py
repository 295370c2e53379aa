"""ctypes front-end of oracle/_ref/libref_thirdparty.so — TEST INFRASTRUCTURE ONLY.

The library is the REFERENCE'S OWN vendored third-party code (tinyobj 1.0.8, stb_image 2.16,
stb_image_resize 0.95) compiled from where it lies under /root/reference by `make oracle-ref`
(oracle/ref_thirdparty.cpp is the C wrapper).  It pins the input side of the hot path: the OBJ/MTL
parse and the texture / cubemap decode.  `available()` is False where the library was never built
(no /root/reference); tests then fall back to the committed fixtures generated from it
(tests/golden/make_ref_golden.py).
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libref_thirdparty.so")
_lib = None


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.ref_tinyobj_load.restype = C.c_void_p
        L.ref_tinyobj_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        for name, ptr in (("ref_stbi_loadf", C.c_float), ("ref_stbi_load", C.c_ubyte)):
            f = getattr(L, name)
            f.restype = C.POINTER(ptr)
            f.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_stbi_failure_reason.restype = C.c_char_p
        L.ref_stbir_resize_float.restype = C.c_int
        L.ref_stbir_resize_float.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int]
        L.ref_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _floats(bits) -> np.ndarray:
    return np.array(bits, dtype=np.uint32).view(np.float32)


def tinyobj_load(obj_path: str, mtl_dir: str):
    """tinyobj::LoadObj as scene.cpp:341 calls it.  Returns (dict | None, error/warning text)."""
    err = C.create_string_buffer(4096)
    p = lib().ref_tinyobj_load(obj_path.encode(), mtl_dir.encode(), err, len(err))
    msg = err.value.decode("utf-8", "replace")
    if not p:
        return None, msg
    try:
        d = json.loads(C.string_at(p).decode("latin-1"))
    finally:
        lib().ref_free(p)
    for k in ("vertices", "normals", "texcoords"):
        d[k] = _floats(d[k])
    for s in d["shapes"]:
        s["indices"] = np.array(s["indices"], dtype=np.int32).reshape(-1, 3)   # (vertex, normal, texcoord) per corner
        s["material_ids"] = np.array(s["material_ids"], dtype=np.int32)
        s["num_face_vertices"] = np.array(s["num_face_vertices"], dtype=np.int32)
    for m in d["materials"]:
        for k in ("ambient", "diffuse", "specular", "transmittance", "emission"):
            m[k] = _floats(m[k])
        for k in ("shininess", "ior", "dissolve"):
            m[k] = _floats(m[k])[0]
    return d, msg


def flatten_like_reference(d):
    """upload_meshes (scene.cpp:202-262) applied to tinyobj's output: per shape, one face per index
    triple; corner attributes gathered by index.  Corners without a normal / texcoord index (-1) make
    the reference read out of bounds; they are returned as NaN here so tests can mask them."""
    meshes = []
    v = d["vertices"].reshape(-1, 3)
    vn = d["normals"].reshape(-1, 3)
    vt = d["texcoords"].reshape(-1, 2)
    for s in d["shapes"]:
        idx = s["indices"]
        n = len(idx) // 3
        idx = idx[: 3 * n].reshape(n, 3, 3)
        pos = v[idx[:, :, 0]]
        nor = np.full((n, 3, 3), np.nan, np.float32)
        uv = np.full((n, 3, 2), np.nan, np.float32)
        has_n = idx[:, :, 1] >= 0
        has_t = idx[:, :, 2] >= 0
        if len(vn):
            nor[has_n] = vn[idx[:, :, 1][has_n]]
        if len(vt):
            uv[has_t] = vt[idx[:, :, 2][has_t]]
        meshes.append(dict(vertices=pos, normals=nor, texcoords=uv, material_ids=s["material_ids"][:n].copy()))
    return meshes


def _image(fn, path, dtype):
    w, h, n = C.c_int(), C.c_int(), C.c_int()
    p = fn(path.encode(), C.byref(w), C.byref(h), C.byref(n))
    if not p:
        return None
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, n.value)).astype(dtype, copy=True)
    finally:
        lib().ref_free(p)


def stbi_loadf(path: str):
    """stbi_loadf(path, STBI_default) -> float32[h, w, c] or None."""
    return _image(lib().ref_stbi_loadf, path, np.float32)


def stbi_load(path: str):
    """stbi_load(path, STBI_default) -> uint8[h, w, c] or None."""
    return _image(lib().ref_stbi_load, path, np.uint8)


def stbir_resize_float(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w, c = img.shape
    out = np.zeros((out_h, out_w, c), np.float32)
    ok = lib().ref_stbir_resize_float(img.ctypes.data_as(C.POINTER(C.c_float)), w, h,
                                      out.ctypes.data_as(C.POINTER(C.c_float)), out_w, out_h, c)
    if not ok:
        raise RuntimeError("stbir_resize_float failed")
    return out
