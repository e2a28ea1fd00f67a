"""Screen-space split of one frame over several GPUs and the gather to rank 0.

The frame is cut into bands of RT_BAND_ROWS rows (counted in output row order); rank r of n
renders the bands b with b % n == r.  Interleaving matters because cost per row is very
uneven (sky rows cost one segment, rows crossing the glass sphere dozens).  Each rank's
bands are packed back to back; rank 0 gathers the packed buffers (torch.distributed gather
-> RCCL over xGMI on GPUs, gloo on CPU) and scatters the rows into place.
"""
from __future__ import annotations

import ctypes

import numpy as np

from .api import RT_BAND_ROWS


def part_bands(height: int, part: int, nparts: int):
    nb = (height + RT_BAND_ROWS - 1) // RT_BAND_ROWS
    return list(range(part, nb, nparts))


def part_row_ranges(height: int, part: int, nparts: int):
    """[(row0, row1), ...] of the output rows rendered by `part`, in packed order."""
    return [(b * RT_BAND_ROWS, min((b + 1) * RT_BAND_ROWS, height)) for b in part_bands(height, part, nparts)]


def part_rows(height: int, part: int, nparts: int) -> int:
    return sum(r1 - r0 for r0, r1 in part_row_ranges(height, part, nparts))


def padded_part_rows(height: int, nparts: int) -> int:
    """Rows of the largest part (part 0): every rank's gather chunk is padded to this."""
    return part_rows(height, 0, nparts)


def assemble_numpy(gathered: np.ndarray, height: int, nparts: int) -> np.ndarray:
    """gathered: (nparts, padded_rows, w, c) packed buffers in part order -> (height, w, c)."""
    out = np.empty((height,) + gathered.shape[2:], dtype=gathered.dtype)
    for part in range(nparts):
        src = 0
        for r0, r1 in part_row_ranges(height, part, nparts):
            out[r0:r1] = gathered[part, src:src + (r1 - r0)]
            src += r1 - r0
    return out


def render_distributed(render_part, height: int, width: int, channels: int, dtype, device, dist):
    """One frame over dist.get_world_size() ranks.

    render_part(part, nparts, out_tensor): fills `out_tensor` (padded_rows, width, channels),
    a torch tensor on `device`, with this rank's packed bands.  Returns the gathered
    (nparts, padded_rows, width, channels) tensor on rank 0, None elsewhere; rank 0 then
    calls assemble (assemble_numpy on CPU, rt_assemble_device on the GPU).
    """
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    rows = padded_part_rows(height, world)
    mine = torch.empty((rows, width, channels), dtype=dtype, device=device)
    render_part(rank, world, mine)
    if world == 1:
        return mine.unsqueeze(0)
    if rank == 0:
        chunks = [torch.empty_like(mine) for _ in range(world)]
        dist.gather(mine, gather_list=chunks, dst=0)
        return torch.stack(chunks, 0)
    dist.gather(mine, gather_list=None, dst=0)
    return None
