"""Pixel-row tile split for multi-GPU rendering (one process per GPU).

The reference is single-GPU (cuda_opengl/src/driver/gpu_info.cpp:26-41).  Pixels are
independent, so the frame shards by surface rows with no data-path exchange; the only
collective is the gather of the finished RGBA8 bands.  Per-pixel RNG seeds are derived
from GLOBAL pixel coordinates (raytrace.cu:227-229 with the full-frame launch geometry),
so an N-way split reproduces the 1-GPU image bit for bit.
"""
from __future__ import annotations

from typing import List, Tuple


def row_bands(height: int, world_size: int, align: int = 1) -> List[Tuple[int, int]]:
    """Contiguous row bands [begin, end) per rank; remainders go to the last ranks.
    Band starts are multiples of `align` (16 keeps bands on the kernel's tile grid)."""
    if world_size < 1 or height < 0:
        raise ValueError("bad split")
    units = (height + align - 1) // align
    base, extra = divmod(units, world_size)
    bands, start = [], 0
    for r in range(world_size):
        n = base + (1 if r >= world_size - extra else 0)
        b, e = min(start * align, height), min((start + n) * align, height)
        bands.append((b, e))
        start += n
    return bands


def band_of_rank(height: int, world_size: int, rank: int, align: int = 1) -> Tuple[int, int]:
    return row_bands(height, world_size, align)[rank]
