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


class BandGather:
    """One all-gather per frame of the finished RGBA8 row bands (torch.distributed: RCCL over
    xGMI on GPUs, gloo in the CPU tests).  Buffers are allocated once; bands that differ by a
    row are padded to the tallest band so a single fixed-size collective suffices."""

    def __init__(self, height: int, width: int, world_size: int, rank: int, device, align: int = 1):
        import torch
        self.bands = row_bands(height, world_size, align)
        self.rank, self.world = rank, world_size
        self.max_rows = max(e - b for b, e in self.bands)
        self.send = torch.zeros((self.max_rows, width, 4), dtype=torch.uint8, device=device)
        self.recv = torch.zeros((world_size * self.max_rows, width, 4), dtype=torch.uint8, device=device)

    def gather(self, band_surface, group=None):
        """band_surface: uint8[rows_of_this_rank, W, 4].  Returns the padded gather buffer
        (asynchronous with respect to the host on GPU streams)."""
        import torch.distributed as dist
        b, e = self.bands[self.rank]
        self.send[: e - b].copy_(band_surface)
        dist.all_gather_into_tensor(self.recv, self.send, group=group)
        return self.recv

    def assemble(self):
        """Full frame uint8[H, W, 4] from the last gather (drops the padding rows)."""
        import torch
        parts = [self.recv[r * self.max_rows: r * self.max_rows + (e - b)] for r, (b, e) in enumerate(self.bands)]
        return torch.cat(parts, dim=0)
