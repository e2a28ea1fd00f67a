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


class BandPipeline:
    """Frames in flight over all ranks: rank r renders its bands of frame k, gathers them to
    rank 0, which scatters the bands into the final image; with two streams frame k+1 renders
    while frame k is in transit (one buffer set per stream).  The only collective on the data path
    is the gather; there is no exchange during rendering.

    render_part(part, nparts, out)   fills the (padded_rows, w, c) tensor `out` with this
                                     rank's packed bands (asynchronously on the *current*
                                     torch stream when `out` lives on a GPU)
    assemble(gathered, image)        rank 0 only: (nparts, padded_rows, w, c) -> (h, w, ci)

    `channels` is what travels between ranks, `image_channels` (default: the same) what the
    assembled frame holds: packed RGB8 parts (3) assembled into an RGBA8 frame (4) move a
    quarter less over the links, which is what bounds a cheap frame.  A world of one renders
    straight into the frame, so render_part then receives an image_channels tensor.
    """

    def __init__(self, height, width, channels, dtype, device, dist, render_part, assemble,
                 always_gather=False, image_channels=None, streams=None):
        import torch
        self.torch, self.dist = torch, dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.h, self.w = height, width
        self.render_part, self.assemble = render_part, assemble
        # a world of one renders straight into the image unless the caller wants the gather
        # path exercised (tests)
        self.direct = self.world == 1 and not (always_gather and dist is not None)
        # S GPU streams, frame k on streams[k % S] with buffer set k % S: the last waves of a part
        # (a pixel's ray tree is one lane's sequential work, ~0.1 ms for the deepest pixels of the
        # default scene, however small the part) then overlap the next frames' parts instead of
        # leaving the GPU idle.  Everything that touches the buffers of frame k (render, gather,
        # assemble) is issued on streams[k % S], so frame k+S is ordered behind it by stream
        # order alone.  (A process gets GPU_MAX_HW_QUEUES hardware queues; streams beyond that
        # share queues and stop overlapping.)
        self.streams = list(streams) if streams else None
        self.nbuf = nbuf = len(self.streams) if self.streams else 2
        rows = padded_part_rows(height, self.world)
        self.mine = [torch.zeros((rows, width, channels), dtype=dtype, device=device) for _ in range(nbuf)]
        nimg = nbuf if (self.streams and not self.direct) else 1
        self.images = ([torch.empty((height, width, image_channels or channels), dtype=dtype, device=device)
                        for _ in range(nimg)] if self.rank == 0 else None)
        self.last_image = 0
        if self.rank == 0 and not self.direct:
            # the gather delivers straight into the slices of one contiguous buffer per frame in flight
            self.gathered = [torch.empty((self.world, rows, width, channels), dtype=dtype, device=device) for _ in range(nbuf)]
            self.chunks = [[g[i] for i in range(self.world)] for g in self.gathered]
        else:
            self.chunks, self.gathered = None, None
        self.k = 0
        self.cur = 0

    @property
    def image(self):
        """The buffer of the most recently ENQUEUED frame (rank 0; None elsewhere): its contents are
        valid once flush() has been called and the current stream has been synchronised."""
        return self.images[self.last_image] if self.images is not None else None

    def use_stream(self, k):
        # self.cur: index into `streams` of the stream the callbacks are to launch on
        # (torch.cuda.current_stream() costs ~9 us a call, a table lookup by the caller nothing)
        self.cur = k % self.nbuf
        if self.streams is not None:
            self.torch.cuda.set_stream(self.streams[self.cur])

    def step(self):
        """Render, gather and (rank 0) assemble one frame; on a GPU everything is only enqueued."""
        k = self.k
        self.k += 1
        if self.direct:
            self.render_part(0, 1, self.images[0])
            return
        b = k % self.nbuf
        buf = self.mine[b]
        self.use_stream(k)
        self.render_part(self.rank, self.world, buf)
        # A blocking-style collective: with RCCL it is enqueued like a kernel (the current stream
        # waits for it on the GPU, the host does not), and it costs the host a fifth of an
        # async_op=True call with a deferred wait().  Frame k+1 renders on the other stream
        # meanwhile.
        self.dist.gather(buf, gather_list=self.chunks[b] if self.rank == 0 else None, dst=0)
        if self.rank == 0:
            self.last_image = k % len(self.images)
            self.assemble(self.gathered[b], self.images[self.last_image])

    def flush(self):
        """Leave the caller on streams[0], ordered behind everything enqueued so far."""
        if self.streams is not None:
            self.cur = 0
            self.torch.cuda.set_stream(self.streams[0])
            for other in self.streams[1:]:
                self.streams[0].wait_stream(other)
