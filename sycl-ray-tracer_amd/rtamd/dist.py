"""Multi-GPU frame assembly: one process per GPU, each renders its interleaved 8-row strips of the SAME
frame; the compact per-rank strip buffers are gathered to rank 0 (torch.distributed: RCCL over xGMI when the
backend is "nccl", gloo in the CPU tests) and de-interleaved into the full frame there.

The reference is single-device (src/app.hpp:43-55); this exchange step is the only collective of the
path. Pixels never interact, so there is nothing to reduce: a gather of N/G pixels per GPU is all that moves
(33 MB of fp32 RGBA per 1080p frame in total)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def strip_rows_of(rank: int, world: int, height: int, strip_rows: int = 8) -> list[int]:
    """Global row indices owned by `rank` (strip k -> rank k % world), top to bottom."""
    return [y for y in range(height) if (y // strip_rows) % world == rank]


def max_local_rows(world: int, height: int, strip_rows: int = 8) -> int:
    return max(len(strip_rows_of(r, world, height, strip_rows)) for r in range(world))


class FrameGatherer:
    """Pre-allocates everything the per-frame exchange needs, so the timed region only launches the
    collective and ONE de-interleave gather."""

    def __init__(self, rank: int, world: int, width: int, height: int, strip_rows: int, device: torch.device,
                 channels: int = 4, dtype=torch.float32, collective: bool | None = None):
        self.rank, self.world, self.height = rank, world, height
        self.collective = world > 1 if collective is None else bool(collective)  # True with one rank: rehearsal
        self.rows = len(strip_rows_of(rank, world, height, strip_rows))
        pad_rows = max_local_rows(world, height, strip_rows)
        # every rank contributes the same (padded) shape: torch.distributed.gather needs equal sizes
        self.local = torch.zeros((pad_rows, width, channels), dtype=dtype, device=device)
        # the reference's RGBA-unorm8 image of the tile (src/render_megakernel.cpp:154-158 writes it inside the timed frame): rendered into,
        # not exchanged — the fp32 plane is what the ranks gather
        self.local_u8 = torch.zeros((pad_rows, width, channels), dtype=torch.uint8, device=device)
        self.frame = self.gathered = self.parts = self.src_row = None
        self._work, self._pending = None, False
        if rank == 0:
            self.frame = torch.zeros((height, width, channels), dtype=dtype, device=device)
            # the ranks' padded tiles land back to back in ONE buffer (the gather list is its slices: no copies) ...
            self.gathered = torch.zeros((world * pad_rows, width, channels), dtype=dtype, device=device)
            self.parts = [self.gathered[r * pad_rows:(r + 1) * pad_rows] for r in range(world)]
            # ... and one index_select over rows (global row y <- row src_row[y] of the buffer) de-interleaves the whole frame
            src = torch.empty(height, dtype=torch.long)
            for r in range(world):
                rows = strip_rows_of(r, world, height, strip_rows)
                src[rows] = r * pad_rows + torch.arange(len(rows))
            self.src_row = src.to(device)

    def gather_begin(self):
        """Starts collecting `self.local` of every rank into `self.frame` on rank 0 and returns at once: the collective is enqueued
        (async_op), nothing waits on the host. The caller has made sure the renderer finished writing `self.local` (end_frame()). Until
        gather_end() the buffer belongs to the exchange: the NEXT frame renders into another FrameGatherer's buffer meanwhile — that is
        how rank k's next frame overlaps the exchange of this one."""
        assert self._work is None and not self._pending, "gather_end() first"
        if self.collective:
            self._work = dist.gather(self.local, self.parts, dst=0, async_op=True)
        elif self.rank == 0:
            self.parts[0].copy_(self.local, non_blocking=True)
        self._pending = True

    def gather_end(self):
        """Completes the exchange begun by gather_begin(): the de-interleave (ONE index_select over rows) on rank 0, then a wait for the
        device to have read `self.local` and written the frame. Returns the frame (rank 0) or None. A no-op without a pending exchange."""
        if not self._pending:
            return self.frame
        if self._work is not None:
            self._work.wait()  # (nccl: orders the current stream behind the collective; gloo: waits)
            self._work = None
        if self.rank == 0:
            torch.index_select(self.gathered, 0, self.src_row, out=self.frame)
        # The renderer writes `self.local` from its own HIP stream: its next frame into this buffer must not start before the
        # exchange has read it (and rank 0's frame is complete when this returns).
        if self.local.is_cuda:
            torch.cuda.current_stream(self.local.device).synchronize()
        self._pending = False
        return self.frame

    def gather(self):
        """Collects `self.local` of every rank into `self.frame` on rank 0. Returns the frame (rank 0) or None."""
        self.gather_begin()
        return self.gather_end()
