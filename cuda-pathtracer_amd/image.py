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


def load_ppm(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        magic, w, h, mx = f.readline().split()
        assert magic == b"P6" and mx == b"255"
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(int(h), int(w), 3)
