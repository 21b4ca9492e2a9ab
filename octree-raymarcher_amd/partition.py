"""Image partition across ranks: 8-row bands dealt round-robin, one gather to rank 0.

Pure host logic, shared by bench.py (RCCL) and the gloo CPU tests.  Rays are independent and the world
is read-only, so every rank holds the whole world and traces bands  rank, rank+N, rank+2N, ...
(svo_trace_rows(band0=rank, band_stride=N, nbands, band_height)); the only exchange step is the gather
of the per-rank G-buffer bands, after which rank 0 de-interleaves them into the frame.
"""
from __future__ import annotations

BAND = 8            # rows per band == tile height of the stack kernel


def bands_per_rank(height: int, world_size: int, band: int = BAND) -> int:
    """Bands every rank traces (the same count on every rank; trailing ones may be padding below the image)."""
    total = (height + band - 1) // band
    return (total + world_size - 1) // world_size


def band_rows(rank: int, world_size: int, k: int, band: int = BAND) -> range:
    """Image rows of the k-th band of `rank`."""
    b = rank + k * world_size
    return range(b * band, (b + 1) * band)


def deinterleave(gathered, height: int, band: int = BAND):
    """gathered: list (one per rank) of tensors/arrays shaped [nb, band, width, ...] -> frame [height, width, ...].

    frame[(k*N + r)*band + j] = gathered[r][k][j]
    """
    n = len(gathered)
    first = gathered[0]
    nb = first.shape[0]
    if hasattr(first, "new_empty"):                       # torch
        import torch
        g = torch.stack(list(gathered), dim=1)            # [nb, N, band, W, ...]
    else:
        import numpy as np
        g = np.stack(list(gathered), axis=1)
    frame = g.reshape((nb * n * band,) + tuple(first.shape[2:]))
    return frame[:height]
