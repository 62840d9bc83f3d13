"""Frame-parallel multi-GPU driver: one process per GPU, frames sharded with no data-path
collective, finished frames gathered to rank 0 (RCCL over xGMI when the backend is "nccl").

The reference has no inference-time multi-GPU path (SURVEY.md §2 "Parallelism"); frames are
independent, so the only exchange is the final gather of ``[n,4,512,512]`` blocks.
The gather of chunk c is issued on a side stream and overlaps the compute of chunk c+1.
Backend-agnostic: the unit tests run it with ``gloo`` on CPU and a stub frame function.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of ``total`` frames for ``rank`` (the first
    ``total % world`` ranks get one extra frame).  Empty slices are allowed (total < world)."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_shard_sizes(total: int, world: int) -> List[int]:
    return [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]


class FrameShardedStream:
    """Pose a stream of ``total`` independent frames split across the ranks of ``group``.

    ``frame_fn(lo, hi) -> Tensor[hi-lo, *frame_shape]`` computes the frames with GLOBAL indices
    [lo, hi) on this rank's device.  ``run()`` returns, with ``gather=True``, the full
    ``[total, *frame_shape]`` tensor on rank 0 (``None`` on the other ranks) with frame i at row i
    regardless of which rank computed it; with ``gather=False`` each rank gets its local block.
    """

    def __init__(self, frame_fn: Callable[[int, int], torch.Tensor], total: int, frame_shape: Sequence[int],
                 dtype: torch.dtype, device: torch.device, chunk: int = 32,
                 group: Optional[dist.ProcessGroup] = None, gather: bool = True):
        self.frame_fn = frame_fn
        self.total = int(total)
        self.frame_shape = tuple(int(s) for s in frame_shape)
        self.dtype = dtype
        self.device = torch.device(device)
        self.chunk = max(1, int(chunk))
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.gather = gather and self.world > 1
        self.lo, self.hi = shard_bounds(self.total, self.rank, self.world)

    def local_range(self) -> Tuple[int, int]:
        return self.lo, self.hi

    def _empty(self, n: int) -> torch.Tensor:
        return torch.empty((n,) + self.frame_shape, dtype=self.dtype, device=self.device)

    def allocate_result(self) -> Optional[torch.Tensor]:
        """The tensor ``run()`` fills (rank 0: all frames; without gather: this rank's block).  Allocating it ahead of
        time keeps a tens-of-GB hipMalloc out of a timed region."""
        if not self.gather:
            return self._empty(self.hi - self.lo)
        return self._empty(self.total) if self.rank == 0 else None

    def run(self, result: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        sizes = all_shard_sizes(self.total, self.world)
        rounds = max((s + self.chunk - 1) // self.chunk for s in sizes)
        if result is None:
            result = self.allocate_result()
        else:
            want = (self.hi - self.lo if not self.gather else self.total,) + self.frame_shape
            if tuple(result.shape) != want or result.dtype != self.dtype or not result.is_contiguous():
                raise RuntimeError(f"result buffer must be a contiguous {self.dtype} tensor of shape {want}")
        cuda = self.device.type == "cuda"
        side = torch.cuda.Stream(device=self.device) if (cuda and self.gather) else None
        dst = dist.get_global_rank(self.group, 0) if (self.gather and self.group is not None) else 0
        keep = []   # receive buffers stay alive until the side stream has drained
        for c in range(rounds):
            a = min(self.lo + c * self.chunk, self.hi)
            b = min(a + self.chunk, self.hi)
            block = self.frame_fn(a, b) if b > a else self._empty(0)
            if tuple(block.shape) != (b - a,) + self.frame_shape:
                raise RuntimeError(f"frame_fn returned {tuple(block.shape)} for frames [{a},{b})")
            if not self.gather:
                result[a - self.lo:b - self.lo] = block
                continue
            # regular collective: every rank sends exactly `chunk` rows; valid counts are analytic
            if block.shape[0] == self.chunk:
                send = block
            else:
                send = torch.zeros((self.chunk,) + self.frame_shape, dtype=self.dtype, device=self.device)
                send[:block.shape[0]] = block
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(self.device))
                send.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else _NullCtx()):
                recv, tails = None, []
                if self.rank == 0:
                    # full chunks land directly in their rows of `result` (no staging copy on the root, which also
                    # has its own frames to compute); only a rank's ragged last chunk goes through a temporary
                    recv = []
                    for r in range(self.world):
                        rlo, rhi = shard_bounds(self.total, r, self.world)
                        ra = min(rlo + c * self.chunk, rhi)
                        rb = min(ra + self.chunk, rhi)
                        if rb - ra == self.chunk:
                            recv.append(result[ra:rb])
                        else:
                            tmp = self._empty(self.chunk)
                            recv.append(tmp)
                            tails.append((tmp, ra, rb))
                dist.gather(send, recv, dst=dst, group=self.group)
                if self.rank == 0:
                    for tmp, ra, rb in tails:
                        if rb > ra:
                            result[ra:rb] = tmp[:rb - ra]
                    keep.append(tails)
        if side is not None:
            torch.cuda.current_stream(self.device).wait_stream(side)
        return result


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
