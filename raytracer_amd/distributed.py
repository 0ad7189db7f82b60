"""One process per GPU: row partition of the frame and the framebuffer gather.

The frame is split into contiguous blocks of rows, one per rank — the image-space
partition the reference applies to its thread pool (Parallel.For over rows,
/root/reference/MyApplication.cs:408).  Pixels are independent, so the only exchange is
the hand-off of the finished blocks to the presenting rank: one gather per frame, RCCL
(torch.distributed backend "nccl") over xGMI between GPUs, gloo between CPU processes
in the tests.  Receive buffers are row slices of rank 0's frame, so the collective
writes each block where Surface.pixels expects it.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def row_block(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [begin, end) of rank `rank`: blocks differ by at most one row and tile [0, height)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside [0, {world})")
    return (height * rank) // world, (height * (rank + 1)) // world


class FrameGatherer:
    """Owns this rank's block buffer and, on rank 0, the full frame.

    ``block``  int32 [rows_of_this_rank, width] — render into it (``block.data_ptr()``);
    ``frame``  int32 [height, width] on rank 0 after ``gather()``, else None.
    """

    def __init__(self, width: int, height: int, rank: int, world: int, device: torch.device,
                 group: Optional[dist.ProcessGroup] = None):
        self.width, self.height, self.rank, self.world, self.group = width, height, rank, world, group
        blocks = [row_block(height, r, world) for r in range(world)]
        y0, y1 = blocks[rank]
        self.rows = y1 - y0
        self.equal = len({b - a for a, b in blocks}) == 1
        self.frame: Optional[torch.Tensor] = None
        self._recv: Optional[List[torch.Tensor]] = None
        self._blocks = blocks
        if rank == 0:
            self.frame = torch.zeros((height, width), dtype=torch.int32, device=device)
        if world == 1:
            self.block = self.frame
            self._send = self.frame
            return
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
        dist.gather(self._send, self._recv if self.rank == 0 else None, dst=0, group=self.group)
        if self.rank == 0 and not self.equal:
            for (a, b), buf in zip(self._blocks, self._recv):
                self.frame[a:b].copy_(buf[: b - a])
        return self.frame


def render_frame(ctx, camera, width: int, height: int, max_depth: int, gatherer: FrameGatherer,
                 stream: Optional["torch.cuda.Stream"] = None) -> Optional[torch.Tensor]:
    """One distributed Render(): this rank's rows through rt_render_rows_device, then the
    gather.  Returns rank 0's frame tensor (device memory), None on the other ranks."""
    y0, y1 = row_block(height, gatherer.rank, gatherer.world)
    s = stream if stream is not None else torch.cuda.current_stream()
    ctx.render_rows_device(camera, width, height, y0, y1, max_depth, gatherer.block.data_ptr(),
                           s.cuda_stream)
    return gatherer.gather()
