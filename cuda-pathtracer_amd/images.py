"""Image providers for the scene loader.  The reference decodes textures with stb_image's
stbi_loadf (material_loader.cpp:97, gpu_processor.cpp:99): 8-bit channels become floats through
`pow(v / 255, 2.2)` for colour channels and `v / 255` for an alpha channel (stb_image.h: the
ldr-to-hdr conversion).  `pil_image_loader` reproduces that contract on top of PIL's decoder; the
decoders themselves (stb vs libjpeg) may differ by an LSB on JPEG data."""
from __future__ import annotations

import numpy as np


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
