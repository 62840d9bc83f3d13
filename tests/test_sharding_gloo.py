"""N>1 path on CPU: world_size-2 gloo processes running the frame-sharded stream with a stub
frame function (the sharding/gather logic is backend-agnostic; the GPU box runs it over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import tha4_amd  # noqa: F401
from tha4_amd.sharding import FrameShardedStream, all_shard_sizes, shard_bounds


def test_shard_bounds_cover_and_balance():
    for total in (0, 1, 7, 8, 64, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = all_shard_sizes(total, world)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == total
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _frame(i):
    g = torch.Generator().manual_seed(1000 + i)
    return torch.rand(2, 4, 4, generator=g)


def _worker(rank, world, port, total, chunk, gather, q, rgba8=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def frame_fn(lo, hi):
        calls.append((lo, hi))
        blk = torch.stack([_frame(i) for i in range(lo, hi)])
        return _to_u8(blk) if rgba8 else blk

    shape, dtype = ((4, 4, 2), torch.uint8) if rgba8 else ((2, 4, 4), torch.float32)
    s = FrameShardedStream(frame_fn, total, shape, dtype, torch.device("cpu"), chunk=chunk, gather=gather)
    out = s.run(s.allocate_result()) if total % 2 else s.run()      # both entry points: caller-provided / internal buffer
    q.put((rank, s.local_range(), calls, None if out is None else out.clone()))
    dist.barrier()
    dist.destroy_process_group()


def _to_u8(blk):
    """stand-in for the display epilogue: [n,2,4,4] fp32 -> [n,4,4,2] uint8 (HWC bytes), a quarter of the bytes"""
    return (blk.permute(0, 2, 3, 1) * 255.0).to(torch.uint8).contiguous()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("total,chunk", [(11, 4), (8, 8), (1, 4), (5, 1), (16, 4), (3, 8)])
def test_two_rank_gather_reassembles_stream(total, chunk):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, chunk, True, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = res[0][3]
    assert res[1][3] is None
    assert full.shape == (total, 2, 4, 4)
    for i in range(total):
        assert torch.equal(full[i], _frame(i))          # frame i lands at row i whichever rank made it
    # every frame computed exactly once, by the rank that owns it
    for r in (0, 1):
        lo, hi = res[r][1]
        assert (lo, hi) == shard_bounds(total, r, 2)
        done = [i for (a, b) in res[r][2] for i in range(a, b)]
        assert done == list(range(lo, hi))


def test_two_rank_no_gather_keeps_local_blocks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, 3, False, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        lo, hi = res[r][1]
        blk = res[r][3]
        assert blk.shape[0] == hi - lo
        for i in range(lo, hi):
            assert torch.equal(blk[i - lo], _frame(i))


def test_single_process_path():
    out = FrameShardedStream(lambda lo, hi: torch.stack([_frame(i) for i in range(lo, hi)]), 5, (2, 4, 4),
                             torch.float32, torch.device("cpu"), chunk=2).run()
    assert out.shape == (5, 2, 4, 4) and torch.equal(out[4], _frame(4))


def test_preallocated_result_is_validated():
    s = FrameShardedStream(lambda lo, hi: torch.stack([_frame(i) for i in range(lo, hi)]), 5, (2, 4, 4), torch.float32,
                           torch.device("cpu"), chunk=2)
    buf = s.allocate_result()
    out = s.run(buf)
    assert out.data_ptr() == buf.data_ptr() and torch.equal(out[3], _frame(3))
    import pytest
    with pytest.raises(RuntimeError):
        s.run(torch.empty(4, 2, 4, 4))


def test_two_rank_rgba8_gather():
    """Display epilogue BEFORE the exchange: uint8 HWC frames gathered, same row placement."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    total, chunk = 13, 4
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, chunk, True, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = res[0][3]
    assert full.dtype == torch.uint8 and full.shape == (total, 4, 4, 2)
    for i in range(total):
        assert torch.equal(full[i], _to_u8(_frame(i)[None])[0])
