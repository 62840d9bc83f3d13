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

import time

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


def tapered_schedule(frames: int, chunk: int, unit: int = 1, tail_frac: float = 0.05) -> List[int]:
    """Gather-round sizes for a rank that produces ``frames`` frames in calls of ``unit`` frames: full ``chunk``s first, then
    rounds that halve down to the smallest one, whose exchange is the only part of the stream that nothing overlaps (round c's
    gather runs on the side stream under round c + 1's compute; behind the LAST round there is no compute left).  The last round
    is at most ``tail_frac`` of the rank's frames (never less than one call: ``unit``), so a short region - the driver's
    ``--steps 20`` - does not end on a full-chunk exchange (round-4 review: 4 of 20 frames = 20 % of the stream exposed).
    Every size is a multiple of ``unit``; the sizes sum to ``frames`` and never increase."""
    if frames < 0 or chunk < 1 or unit < 1 or frames % unit:
        raise ValueError(f"bad schedule request frames={frames} chunk={chunk} unit={unit}")
    chunk = max(unit, chunk // unit * unit)
    last = max(unit, int(frames * tail_frac) // unit * unit)
    taper, t = [], last                           # last, 2 last, 4 last, ... (all below a full chunk)
    while t < chunk:
        taper.append(t)
        t *= 2
    while taper and sum(taper) > frames:          # a stream shorter than its taper: drop the largest rounds
        taper.pop()
    body = frames - sum(taper)
    sizes = [chunk] * (body // chunk) + ([body % chunk] if body % chunk else []) + taper
    sizes.sort(reverse=True)
    assert sum(sizes) == frames and all(x > 0 and x % unit == 0 for x in sizes)
    return sizes


class FrameShardedStream:
    """Pose a stream of ``total`` independent frames split across the ranks of ``group``.

    ``frame_fn(lo, hi) -> Tensor[hi-lo, *frame_shape]`` computes the frames with GLOBAL indices
    [lo, hi) on this rank's device.  ``run()`` returns, with ``gather=True``, the full
    ``[total, *frame_shape]`` tensor on rank 0 (``None`` on the other ranks) with frame i at row i
    regardless of which rank computed it; with ``gather=False`` each rank gets its local block.
    """

    def __init__(self, frame_fn: Callable[[int, int], torch.Tensor], total: int, frame_shape: Sequence[int],
                 dtype: torch.dtype, device: torch.device, chunk: int = 32,
                 group: Optional[dist.ProcessGroup] = None, gather: bool = True,
                 on_chunk: Optional[Callable[[int, int, torch.Tensor], None]] = None, ring_slots: int = 3,
                 force_collective: bool = False, schedule: Optional[Sequence[int]] = None, record_rounds: bool = False):
        """``on_chunk(lo, hi, frames)`` switches the root from an ARCHIVE of the whole stream (``allocate_result``:
        ``total`` frames on rank 0 - 2000 steps x 32 frames x 8 ranks would be 2 TB) to a STREAM: rank 0 owns a ring of
        ``ring_slots`` buffers of one gather round each (``world x chunk`` frames: 3 x 8 x 32 x 4 MiB = 3 GiB) and hands
        every arrived span - global frame ids [lo, hi), a view into the ring - to the consumer, which is what a real
        receiver (encoder, compositor, network sender) looks like.  The callback runs with the gather's side stream
        current, i.e. whatever it enqueues there is ordered after the arrival of the data and before the slot is
        overwritten ``ring_slots`` rounds later; a consumer that works on another stream must make that stream wait on
        an event it records in the callback and must be done before the slot comes round again.  ``run()`` then
        returns None.  Without an exchange - one process, no process group, or ``gather=False`` - the consumer is served
        just the same: every block this rank produces goes to ``on_chunk`` as it is finished (on the current stream; the
        block is the frame function's own tensor), nothing is archived and ``run()`` returns None on every rank.
        ``force_collective`` (tests) takes the gather path - side stream, ``dist.gather`` into views of the
        destination - even in a one-rank group: the only way a 1-GPU box executes RCCL at all (it refuses two ranks on one
        device)."""
        self.frame_fn = frame_fn
        # record_rounds: time every gather round where it executes (events on the side stream; wall clock on CPU tensors) for round_report() -
        # first-contact diagnostics of an exchange nobody has measured on N > 1 GPUs
        self.record_rounds = bool(record_rounds)
        self._round_marks = []
        self.on_chunk = on_chunk
        self.ring_slots = max(2, int(ring_slots))
        self.total = int(total)
        self.frame_shape = tuple(int(s) for s in frame_shape)
        self.dtype = dtype
        self.device = torch.device(device)
        self.chunk = max(1, int(chunk))
        self.schedule = [int(x) for x in schedule] if schedule is not None else None
        if self.schedule is not None:
            if any(x < 1 for x in self.schedule):
                raise ValueError("schedule entries must be positive")
            self.chunk = max(self.schedule) if self.schedule else self.chunk
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.gather = gather and (self.world > 1 or (force_collective and self.distributed))
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

    def ring_bytes(self) -> int:
        """Bytes rank 0 holds in streaming mode (``on_chunk``): the whole receive side of the exchange."""
        n = 1
        for d in self.frame_shape:
            n *= d
        return self.ring_slots * self.world * self.chunk * n * torch.empty((), dtype=self.dtype).element_size()

    def round_report(self) -> list:
        """Per gather round of the last ``run(record_rounds=True)``: frames this rank contributed / all ranks delivered, whether it was
        one collective (``full``) or exact-size point-to-point transfers, and the milliseconds the exchange (+ the consumer callback)
        took on this rank's side stream.  Synchronises the device.  On rank 0 ``GBps`` is the rate of bytes ARRIVING from the other ranks."""
        if self.device.type == "cuda" and self._round_marks:
            torch.cuda.synchronize(self.device)
        n = 1
        for d in self.frame_shape:
            n *= d
        fb = n * torch.empty((), dtype=self.dtype).element_size()
        out = []
        for m in self._round_marks:
            ms = m["t0"].elapsed_time(m["t1"]) if m.get("cuda") else 1e3 * (m["t1"] - m["t0"])
            arriving = (m["all"] - m["own"]) * fb
            out.append({"round": m["round"], "frames_this_rank": m["own"], "frames_all_ranks": m["all"], "full": m["full"], "ms": round(ms, 4),
                        "GBps": round(arriving / max(ms, 1e-6) / 1e6, 2) if self.rank == 0 else round(m["own"] * fb / max(ms, 1e-6) / 1e6, 2)})
        return out

    def run(self, result: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        self._round_marks = []
        sizes = all_shard_sizes(self.total, self.world)
        if self.schedule is not None:
            if sum(self.schedule) < max(sizes):
                raise RuntimeError(f"the gather schedule covers {sum(self.schedule)} frames, the largest shard has {max(sizes)}")
            starts, acc = [], 0
            for x in self.schedule:
                starts.append(acc)
                acc += x
            rounds = sum(1 for st in starts if st < max(sizes))
            round_size = list(self.schedule)
        else:
            rounds = max((s + self.chunk - 1) // self.chunk for s in sizes)
            starts = [c * self.chunk for c in range(rounds)]
            round_size = [self.chunk] * rounds
        streaming = self.gather and self.on_chunk is not None
        local_streaming = (not self.gather) and self.on_chunk is not None      # no exchange: this rank's blocks go straight to the consumer
        ring = None
        if streaming or local_streaming:
            if result is not None:
                raise RuntimeError("a result buffer and on_chunk are exclusive: the stream is consumed, not archived")
            if streaming and self.rank == 0:
                ring = [self._empty(self.world * self.chunk) for _ in range(self.ring_slots)]
        elif result is None:
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
            a = min(rlo + starts[c], rhi)
            return a, min(a + round_size[c], rhi)

        if self.gather and rounds > 0:
            # RCCL builds a group's communicator lazily inside the first operation and that blocks until EVERY rank of the
            # group has joined: a ragged first round in which some rank has no rows (total < world) would leave the others
            # waiting in batch_isend_irecv forever.  One cheap collective that all ranks issue settles it.
            dist.barrier(group=self.group)
        for c in range(rounds):
            a, b = span(self.rank, c)
            block = self.frame_fn(a, b) if b > a else None
            if block is not None and (tuple(block.shape) != (b - a,) + self.frame_shape or block.dtype != self.dtype):
                raise RuntimeError(f"frame_fn returned {tuple(block.shape)} {block.dtype} for frames [{a},{b})")
            if not self.gather:
                if block is not None:
                    if local_streaming:
                        self.on_chunk(a, b, block)
                    else:
                        result[a - self.lo:b - self.lo] = block
                continue
            spans = [span(r, c) for r in range(self.world)]
            full_round = all(rb - ra == round_size[c] for ra, rb in spans)
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(self.device))
                if block is not None:
                    block.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else _NullCtx()):
                mark = None
                if self.record_rounds:
                    mark = {"round": c, "own": b - a, "all": sum(rb - ra for ra, rb in spans), "full": full_round, "cuda": side is not None}
                    if side is not None:
                        mark["t0"] = torch.cuda.Event(enable_timing=True)
                        mark["t0"].record(side)
                    else:
                        mark["t0"] = time.perf_counter()
                if self.rank == 0:
                    if streaming:        # this round's slot of the ring: rank r's rows at [r * chunk, r * chunk + its count)
                        slot = ring[c % self.ring_slots]
                        dest = [slot[r * self.chunk:r * self.chunk + (rb - ra)] for r, (ra, rb) in enumerate(spans)]
                    else:                # archive: every chunk lands directly in its rows of `result`
                        dest = [result[ra:rb] for ra, rb in spans]
                if full_round:
                    # regular round: ONE collective, no staging copy on the root
                    dist.gather(block, dest if self.rank == 0 else None, dst=root, group=self.group)
                else:
                    # ragged round (the tail of the stream): exact-size point-to-point transfers - a rank with a short
                    # or empty tail sends only what it has (nothing is zero-padded to a full chunk)
                    if self.rank == 0:
                        if block is not None:
                            dest[0].copy_(block)
                        ops = [dist.P2POp(dist.irecv, dest[r], dist.get_global_rank(self.group, r) if self.group is not None else r,
                                          group=self.group)
                               for r, (ra, rb) in enumerate(spans) if r != 0 and rb > ra]
                    else:
                        ops = [dist.P2POp(dist.isend, block, root, group=self.group)] if block is not None else []
                    if ops:
                        for req in dist.batch_isend_irecv(ops):
                            req.wait()
                if streaming and self.rank == 0:
                    for r, (ra, rb) in enumerate(spans):
                        if rb > ra:
                            self.on_chunk(ra, rb, dest[r])
                if mark is not None:
                    if side is not None:
                        mark["t1"] = torch.cuda.Event(enable_timing=True)
                        mark["t1"].record(side)
                    else:
                        mark["t1"] = time.perf_counter()
                    self._round_marks.append(mark)
        if side is not None:
            torch.cuda.current_stream(self.device).wait_stream(side)
        return result


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
