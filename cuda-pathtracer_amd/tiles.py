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


def interleaved_bands(height: int, world_size: int, rank: int, band_rows: int = 16) -> List[Tuple[int, int]]:
    """Interleaved assignment: the frame is cut into bands of `band_rows` rows (the last one may be
    shorter) and band j belongs to rank j % world_size.  Returns the frame-row ranges [begin, end) of `rank`, in the
    order its band-local buffers hold them.  Measured on MI355X (profiles/r02_band_proxy_contiguous.json): contiguous
    eighths of the headline frame differ by up to 13 % in cost (the middle of the picture is more expensive than its
    top), interleaved 16-row bands level that out."""
    if world_size < 1 or not (0 <= rank < world_size) or band_rows < 1:
        raise ValueError("bad split")
    return [(y, min(y + band_rows, height)) for y in range(rank * band_rows, height, world_size * band_rows)]


class BandGather:
    """One all-gather per frame of the finished RGBA8 row bands (torch.distributed: RCCL over
    xGMI on GPUs, gloo in the CPU tests).  Buffers are allocated once; bands that differ by a
    row are padded to the tallest band so a single fixed-size collective suffices.
    interleave = band_rows > 0: ranks own interleaved bands (interleaved_bands) instead of one contiguous band each."""

    def __init__(self, height: int, width: int, world_size: int, rank: int, device, align: int = 1, interleave: int = 0):
        import torch
        self.interleave = interleave
        self.rank, self.world = rank, world_size
        if interleave:
            per_rank = [interleaved_bands(height, world_size, r, interleave) for r in range(world_size)]
            self.rows_of = [sum(e - b for b, e in bands) for bands in per_rank]
            self.bands = [(0, n) for n in self.rows_of]          # a rank's local buffer: rows [0, rows_of[rank])
            self.max_rows = max(self.rows_of)
            # frame row -> row of the padded gather buffer
            index = torch.empty(height, dtype=torch.long)
            for r, bands in enumerate(per_rank):
                local = 0
                for b, e in bands:
                    index[b:e] = torch.arange(r * self.max_rows + local, r * self.max_rows + local + (e - b))
                    local += e - b
            self.index = index.to(device)
            self.send = torch.zeros((self.max_rows, width, 4), dtype=torch.uint8, device=device)
            self.recv = torch.zeros((world_size * self.max_rows, width, 4), dtype=torch.uint8, device=device)
            return
        self.bands = row_bands(height, world_size, align)
        self.max_rows = max(e - b for b, e in self.bands)
        self.send = torch.zeros((self.max_rows, width, 4), dtype=torch.uint8, device=device)
        self.recv = torch.zeros((world_size * self.max_rows, width, 4), dtype=torch.uint8, device=device)

    def send_rows(self):
        """This rank's part of the send buffer, uint8[rows_of_this_rank, W, 4]: a renderer that writes its surface here
        (FrameRenderer(surface=...)) needs no staging copy before the collective."""
        b, e = self.bands[self.rank]
        return self.send[: e - b]

    def gather(self, band_surface, group=None):
        """band_surface: uint8[rows_of_this_rank, W, 4] (send_rows() itself: no copy).  Returns the padded gather buffer
        (asynchronous with respect to the host on GPU streams)."""
        import torch.distributed as dist
        b, e = self.bands[self.rank]
        if band_surface.data_ptr() != self.send.data_ptr():
            self.send[: e - b].copy_(band_surface)
        dist.all_gather_into_tensor(self.recv, self.send, group=group)
        return self.recv

    def assemble(self):
        """Full frame uint8[H, W, 4] from the last gather (drops the padding rows)."""
        import torch
        if self.interleave:
            return self.recv.index_select(0, self.index)
        parts = [self.recv[r * self.max_rows: r * self.max_rows + (e - b)] for r, (b, e) in enumerate(self.bands)]
        return torch.cat(parts, dim=0)


def rank_times_ms(seconds: float, device, group=None):
    """Every rank's own time for the timed region, gathered on all ranks: (max, mean, list) in milliseconds.  bench.py
    reports max and mean next to the job's time (per-rank times; the gap between them is load imbalance)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([seconds * 1e3], dtype=torch.float64, device=device)
    every = torch.zeros(world, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(every, mine, group=group)
    vals = [float(v) for v in every.tolist()]
    return max(vals), sum(vals) / len(vals), vals


def time_gather_ms(bg: "BandGather", reps: int, synchronize, group=None) -> float:
    """Mean time of ONE all-gather of the RGBA8 bands with nothing else in flight (the gather reported
    separately): `reps` gathers between two barriers, the slowest rank's time.  synchronize(): waits for the device
    (torch.cuda.synchronize on GPUs, a no-op for gloo)."""
    import time
    import torch
    import torch.distributed as dist
    bg.gather(bg.send_rows(), group=group)        # communicator / buffers warm
    synchronize()
    dist.barrier(group=group)
    t0 = time.perf_counter()
    for _ in range(reps):
        bg.gather(bg.send_rows(), group=group)
    synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=bg.send.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item()) / max(reps, 1) * 1e3
