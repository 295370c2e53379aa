"""Image output for the RGBA8 surface (row 0 = top).  The reference only ever shows its surface in
a GL window (driver/interop.cpp:58-72); a headless renderer needs a file writer."""
from __future__ import annotations

import numpy as np


def save_ppm(path: str, rgba: np.ndarray) -> None:
    """rgba: uint8[H, W, 4] (alpha ignored) -> binary PPM (P6)."""
    rgba = np.asarray(rgba)
    if rgba.dtype != np.uint8 or rgba.ndim != 3 or rgba.shape[2] < 3:
        raise ValueError("expected uint8[H, W, >=3]")
    h, w = rgba.shape[:2]
    with open(path, "wb") as f:
        f.write(f"P6 {w} {h} 255\n".encode())
        f.write(np.ascontiguousarray(rgba[:, :, :3]).tobytes())


def save_png(path: str, img: np.ndarray, drop_alpha: bool = True) -> None:
    """uint8[H, W] or uint8[H, W, 1..4] -> PNG through libptamd (ptamd_image_save_png).  The render surface's alpha is 0
    (raytrace.cu:232), so by default a 4-channel image is written as RGB."""
    import ctypes as C
    from . import native as N
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim not in (2, 3):
        raise ValueError("expected uint8[H, W] or uint8[H, W, C]")
    if img.ndim == 2:
        img = img[:, :, None]
    if img.shape[2] == 4 and drop_alpha:
        img = img[:, :, :3]
    img = np.ascontiguousarray(img)
    h, w, c = img.shape
    N.check(N.load().ptamd_image_save_png(path.encode(), img.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, c))


def load_ppm(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        magic, w, h, mx = f.readline().split()
        assert magic == b"P6" and mx == b"255"
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(int(h), int(w), 3)
