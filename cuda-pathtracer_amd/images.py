"""Image decoding for the scene loader.  The reference decodes textures with stb_image's
stbi_loadf (material_loader.cpp:97, gpu_processor.cpp:99): 8-bit channels become floats through
`pow(v / 255, 2.2)` for colour channels and `v / 255` for an alpha channel (stb_image.h: the
ldr-to-hdr conversion).

`load_image` / `load_image8` call libptamd's built-in decoders — JPEG (host/image_decode.cpp), PNG
(host/image_png.cpp) and Radiance HDR —, whose pixels are bit-identical to stb_image 2.16's
(tests/test_ref_thirdparty.py); they are what `HostScene.load` uses by default.
`pil_image_loader` is an optional provider for formats the built-in decoders do not read (BMP, TGA,
GIF ...): it reproduces the float contract on top of PIL, but PIL's JPEG decoder (libjpeg) differs
from stb's by an LSB here and there, so it is not used for parity runs."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import native as N


def load_image(path: str) -> np.ndarray:
    """stbi_loadf(path, STBI_default): float32[h, w, c], c = 1 (grayscale file) or 3.  Raises on failure."""
    w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
    data = C.POINTER(C.c_float)()
    lib = N.load()
    N.check(lib.ptamd_image_loadf(path.encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(data)))
    try:
        return np.ctypeslib.as_array(data, shape=(h.value, w.value, c.value)).copy()
    finally:
        lib.ptamd_image_free(data)


def load_image8(path: str) -> np.ndarray:
    """stbi_load(path, STBI_default): uint8[h, w, c]."""
    w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
    data = C.POINTER(C.c_uint8)()
    lib = N.load()
    N.check(lib.ptamd_image_load8(path.encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(data)))
    try:
        return np.ctypeslib.as_array(data, shape=(h.value, w.value, c.value)).copy()
    finally:
        lib.ptamd_image_free(data)


def resize_float(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """stbir_resize_float with its defaults: float32[h, w, c] -> float32[out_h, out_w, c]."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    out = np.zeros((out_h, out_w, c), np.float32)
    N.check(N.load().ptamd_image_resize_float(img.ctypes.data_as(C.POINTER(C.c_float)), w, h,
                                              out.ctypes.data_as(C.POINTER(C.c_float)), out_w, out_h, c))
    return out


def native_image_loader(path: str):
    """`f(path) -> float32[h, w, c] | None` over the built-in decoder (for cubemap_for_scene)."""
    try:
        return load_image(path)
    except (N.PtamdError, OSError):
        return None


def ldr_to_float(img8: np.ndarray) -> np.ndarray:
    """uint8[h, w, c] -> float32[h, w, c] with stbi_loadf's gamma rule (alpha = last channel when c is even)."""
    img8 = np.asarray(img8, dtype=np.uint8)
    if img8.ndim == 2:
        img8 = img8[:, :, None]
    c = img8.shape[2]
    n_colour = c if (c & 1) else c - 1
    out = np.empty(img8.shape, dtype=np.float32)
    base = img8[:, :, :n_colour].astype(np.float32) / np.float32(255.0)
    out[:, :, :n_colour] = np.power(base, np.float32(2.2), dtype=np.float32)
    if n_colour < c:
        out[:, :, n_colour] = img8[:, :, n_colour].astype(np.float32) / np.float32(255.0)
    return out


def pil_image_loader(path: str):
    """`f(path) -> float32[h, w, c] | None` for HostScene.load / cubemap_for_scene."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            if im.mode not in ("L", "LA", "RGB", "RGBA"):
                im = im.convert("RGB")
            return ldr_to_float(np.asarray(im))
    except (OSError, ImportError):
        return None
