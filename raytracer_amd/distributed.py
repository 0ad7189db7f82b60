"""One process per GPU: partition of the frame's rows and the framebuffer gather.

The frame is cut into bands of BAND_ROWS rows and rank r of G takes the bands r, r + G,
r + 2G, ... — an image-space partition like the one the reference applies to its thread pool
(Parallel.For over rows, /root/reference/MyApplication.cs:408), interleaved because rows cost
very different amounts (mirror spheres, the reflecting ceiling) and the reference's pool
balances them dynamically while ranks cannot.  Pixels are independent, so the only exchange
is the hand-off of the finished blocks to the presenting rank: one gather per frame, RCCL
(torch.distributed backend "nccl") over xGMI between GPUs, gloo between CPU processes in the
tests, followed on rank 0 by one row permutation that puts the bands where Surface.pixels
expects them.  ``interleave=False`` gives contiguous blocks, whose receive buffers are row
slices of rank 0's frame (no permutation).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

BAND_ROWS = 16   # one wave of the primary kernel covers 16 x 16 pixels


def row_block(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [begin, end) of rank `rank`: blocks differ by at most one row and tile [0, height)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside [0, {world})")
    return (height * rank) // world, (height * (rank + 1)) // world


def band_rows_of(height: int, rank: int, world: int, band_rows: int = BAND_ROWS) -> List[int]:
    """Frame rows of rank `rank` in the interleaved partition, in the order of its block."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside [0, {world})")
    rows: List[int] = []
    bands = (height + band_rows - 1) // band_rows
    for b in range(rank, bands, world):
        rows.extend(range(b * band_rows, min(height, (b + 1) * band_rows)))
    return rows


class FrameGatherer:
    """Owns this rank's block buffer and, on rank 0, the full frame.

    ``block``  int32 [rows_of_this_rank, width] — render into it (``block.data_ptr()``);
    ``frame``  int32 [height, width] on rank 0 after ``gather()``, else None.
    """

    def __init__(self, width: int, height: int, rank: int, world: int, device: torch.device,
                 group: Optional[dist.ProcessGroup] = None, interleave: bool = True,
                 band_rows: int = BAND_ROWS):
        self.width, self.height, self.rank, self.world, self.group = width, height, rank, world, group
        self.band_rows = band_rows
        self.interleave = interleave and world > 1
        self.frame: Optional[torch.Tensor] = None
        self._recv: Optional[List[torch.Tensor]] = None
        self._perm: Optional[torch.Tensor] = None
        # Rendering, the collective and the row permutation are ordered on ONE stream of this
        # gatherer's own (the library's default stream is ordered against nothing).
        self.stream = torch.cuda.Stream(device) if device.type == "cuda" else None
        if rank == 0:
            self.frame = torch.zeros((height, width), dtype=torch.int32, device=device)
        if world == 1:
            self.rows = height
            self.block = self.frame
            self._send = self.frame
            self.equal = True
            return
        if self.interleave:
            mine = [band_rows_of(height, r, world, band_rows) for r in range(world)]
            self.rows = len(mine[rank])
            tall = max(len(m) for m in mine)
            self.equal = False
            self._send = torch.zeros((tall, width), dtype=torch.int32, device=device)
            self.block = self._send[: self.rows]
            if rank == 0:
                stacked = torch.zeros((world, tall, width), dtype=torch.int32, device=device)
                self._stacked = stacked
                self._recv = [stacked[r] for r in range(world)]
                # frame row y comes from row perm[y] of the stacked blocks
                perm = torch.empty(height, dtype=torch.int64)
                for r, m in enumerate(mine):
                    perm[torch.tensor(m, dtype=torch.int64)] = r * tall + torch.arange(len(m), dtype=torch.int64)
                self._perm = perm.to(device)
            return
        blocks = [row_block(height, r, world) for r in range(world)]
        y0, y1 = blocks[rank]
        self.rows = y1 - y0
        self.equal = len({b - a for a, b in blocks}) == 1
        self._blocks = blocks
        if self.equal:
            # Rank 0 renders straight into its slice of the frame; the others into a block.
            self._send = (self.frame[y0:y1] if rank == 0
                          else torch.zeros((self.rows, width), dtype=torch.int32, device=device))
            self.block = self._send
            if rank == 0:
                self._recv = [self.frame[a:b] for a, b in blocks]
        else:
            # Collectives want equal counts: pad every block to the tallest one.
            tall = max(b - a for a, b in blocks)
            self._send = torch.zeros((tall, width), dtype=torch.int32, device=device)
            self.block = self._send[: self.rows]
            if rank == 0:
                self._recv = [torch.zeros((tall, width), dtype=torch.int32, device=device)
                              for _ in blocks]

    def gather(self) -> Optional[torch.Tensor]:
        """Enqueue the gather of all blocks to rank 0 (one collective per frame)."""
        if self.world == 1:
            return self.frame
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self._gather()
        else:
            self._gather()
        return self.frame

    def _gather(self) -> None:
        if self._send.is_cuda and dist.get_backend(self.group) == "gloo":
            self._gather_through_host()
        else:
            dist.gather(self._send, self._recv if self.rank == 0 else None, dst=0, group=self.group)
        if self.rank == 0:
            if self.interleave:
                torch.index_select(self._stacked.view(-1, self.width), 0, self._perm, out=self.frame)
            elif not self.equal:
                for (a, b), buf in zip(self._blocks, self._recv):
                    self.frame[a:b].copy_(buf[: b - a])


    def _gather_through_host(self) -> None:
        """gloo with device tensors (ranks rehearsed on fewer GPUs than ranks): the blocks travel
        through host memory on THIS gatherer's stream.  Handing gloo the device tensors would make
        it copy them on high-priority side streams of its own, and on a GPU that two processes
        share a high-priority stream waiting for a kernel of a normal-priority one pauses both
        processes' queues for seconds at a time (measured with a plain busy-wait kernel as well:
        tools/two_rank_probe.py); host tensors keep gloo off the device."""
        host_send = self._send.cpu()   # on the current stream, behind the frame's kernels; blocking
        host_recv = [torch.empty_like(host_send) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(host_send, host_recv, dst=0, group=self.group)
        if self.rank == 0:
            for dst, src in zip(self._recv, host_recv):
                dst.copy_(src, non_blocking=False)


class FramePipeline:
    """Frames in flight: while frame i is gathered (and permuted on rank 0), frame i + 1 renders.

    ``depth`` FrameGatherers take turns, each with its own buffers and stream.  With ONE render
    context (``render_frame(ctx, ...)``) rendering is kept to one frame at a time, because a
    context has one set of ray queues: a slot's stream waits for the previous frame's kernels
    before it renders, not for the previous frame's collective.  With ``contexts`` (one render
    context per slot, each told with ``set_device_share`` how many there are) the frames of the
    slots also RENDER side by side: a rank's share of a frame split over several GPUs is bound by
    its chain of dependent reflection levels, during which most of the device idles, and a second
    and third frame in flight fill it (measured on one MI355X, one eighth of the 3840x2160 frame:
    0.361 ms per frame with one context, 0.250 with two, 0.231 with three).  Collectives are
    issued in frame order on every rank.  The frame tensor that ``render_frame`` returns on rank
    0 is overwritten ``depth`` frames later.
    """

    def __init__(self, width: int, height: int, rank: int, world: int, device: torch.device,
                 group: Optional[dist.ProcessGroup] = None, interleave: bool = True,
                 band_rows: int = BAND_ROWS, depth: int = 2, contexts: Optional[Sequence] = None):
        if depth < 1:
            raise ValueError("depth must be at least 1")
        if contexts is not None and len(contexts) != depth:
            raise ValueError("one render context per slot")
        self.width, self.height = width, height
        self.contexts = list(contexts) if contexts is not None else None
        self.slots = [FrameGatherer(width, height, rank, world, device, group, interleave, band_rows)
                      for _ in range(depth)]
        self._count = 0
        self._rendered: Optional[torch.cuda.Event] = None

    def begin(self) -> FrameGatherer:
        """The next slot; with one shared context its stream is ordered behind the rendering of the
        previous frame: fill ``slot.block`` on ``slot.stream`` (``render_block``), then call
        ``finish(slot)``."""
        g = self.slots[self._count % len(self.slots)]
        self._count += 1
        if self.contexts is None and g.stream is not None and self._rendered is not None:
            g.stream.wait_event(self._rendered)
        return g

    def finish(self, g: FrameGatherer) -> Optional[torch.Tensor]:
        """Marks the slot's rendering as enqueued and enqueues its gather."""
        if self.contexts is None and g.stream is not None:
            self._rendered = torch.cuda.Event()
            self._rendered.record(g.stream)
        return g.gather()

    def context_of(self, g: FrameGatherer):
        """The render context that belongs to a slot (pipelines built with ``contexts``)."""
        return self.contexts[self.slots.index(g)]

    def render_frame(self, ctx, camera, max_depth: int) -> Optional[torch.Tensor]:
        """One frame through the next slot; ``ctx`` is ignored by pipelines built with ``contexts``."""
        g = self.begin()
        render_block(self.context_of(g) if self.contexts is not None else ctx, camera, self.width, self.height,
                     max_depth, g)
        return self.finish(g)

    def synchronize(self) -> None:
        for g in self.slots:
            if g.stream is not None:
                g.stream.synchronize()


def render_block(ctx, camera, width: int, height: int, max_depth: int, gatherer: FrameGatherer) -> None:
    """Enqueue this rank's part of the frame into ``gatherer.block`` (device memory) on the
    gatherer's stream."""
    g = gatherer
    stream_handle = g.stream.cuda_stream if g.stream is not None else 0
    if g.interleave:
        ctx.render_bands_device(camera, width, height, g.band_rows, g.rank, g.world, max_depth,
                                g.block.data_ptr(), stream_handle)
    else:
        y0, y1 = row_block(height, g.rank, g.world)
        ctx.render_rows_device(camera, width, height, y0, y1, max_depth, g.block.data_ptr(), stream_handle)


def render_frame(ctx, camera, width: int, height: int, max_depth: int,
                 gatherer: FrameGatherer) -> Optional[torch.Tensor]:
    """One distributed Render(): this rank's rows, then the gather, both enqueued on
    ``gatherer.stream`` (synchronise it, or make your stream wait for it, before reading).
    Returns rank 0's frame tensor (device memory), None on the other ranks."""
    render_block(ctx, camera, width, height, max_depth, gatherer)
    return gatherer.gather()
