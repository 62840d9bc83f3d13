"""Frame-parallel multi-GPU driver: one process per GPU, frames sharded with no data-path
collective, finished frames gathered to rank 0 (RCCL over xGMI when the backend is "nccl").

The reference has no inference-time multi-GPU path (SURVEY.md §2 "Parallelism"); frames are
independent, so the only exchange is the final gather of ``[n,4,512,512]`` blocks.
The gather of chunk c is issued on a side stream and overlaps the compute of chunk c+1.  A "frame" is whatever
``frame_fn`` produces per index: fp32 ``[4,512,512]`` posed frames, or - with the display epilogue applied before the
exchange - ``uint8 [512,512,4]`` (a quarter of the bytes).
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
        root = dist.get_global_rank(self.group, 0) if (self.gather and self.group is not None) else 0

        def span(r, c):                      # global rows [a, b) rank r produces in round c (analytic: no size exchange)
            rlo, rhi = shard_bounds(self.total, r, self.world)
            a = min(rlo + c * self.chunk, rhi)
            return a, min(a + self.chunk, rhi)

        for c in range(rounds):
            a, b = span(self.rank, c)
            block = self.frame_fn(a, b) if b > a else None
            if block is not None and (tuple(block.shape) != (b - a,) + self.frame_shape or block.dtype != self.dtype):
                raise RuntimeError(f"frame_fn returned {tuple(block.shape)} {block.dtype} for frames [{a},{b})")
            if not self.gather:
                if block is not None:
                    result[a - self.lo:b - self.lo] = block
                continue
            spans = [span(r, c) for r in range(self.world)]
            full_round = all(rb - ra == self.chunk for ra, rb in spans)
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(self.device))
                if block is not None:
                    block.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else _NullCtx()):
                if full_round:
                    # regular round: ONE collective, every chunk lands directly in its rows of `result` on the root
                    recv = [result[ra:rb] for ra, rb in spans] if self.rank == 0 else None
                    dist.gather(block, recv, dst=root, group=self.group)
                else:
                    # ragged round (the tail of the stream): exact-size point-to-point transfers - a rank with a short
                    # or empty tail sends only what it has (nothing is zero-padded to a full chunk)
                    if self.rank == 0:
                        if block is not None:
                            result[a:b] = block
                        ops = [dist.P2POp(dist.irecv, result[ra:rb], dist.get_global_rank(self.group, r) if self.group is not None else r,
                                          group=self.group)
                               for r, (ra, rb) in enumerate(spans) if r != 0 and rb > ra]
                    else:
                        ops = [dist.P2POp(dist.isend, block, root, group=self.group)] if block is not None else []
                    if ops:
                        for req in dist.batch_isend_irecv(ops):
                            req.wait()
        if side is not None:
            torch.cuda.current_stream(self.device).wait_stream(side)
        return result


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
