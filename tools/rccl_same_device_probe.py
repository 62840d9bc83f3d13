#!/usr/bin/env python3
"""Can two ranks share ONE GPU under RCCL?  (GPU box probe: the 1-GPU boxes never execute the RCCL gather otherwise.)"""
import os, sys, socket, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        import tha4_amd  # noqa
        from tha4_amd.sharding import FrameShardedStream
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        def frame_fn(lo, hi):
            return torch.stack([torch.full((2, 8, 8), float(i), device=dev) for i in range(lo, hi)])
        out = FrameShardedStream(frame_fn, 11, (2, 8, 8), torch.float32, dev, chunk=4, gather=True).run()
        torch.cuda.synchronize()
        ok = True
        if rank == 0:
            ok = all(float(out[i, 0, 0, 0]) == i for i in range(11))
        seen = []
        FrameShardedStream(frame_fn, 23, (2, 8, 8), torch.float32, dev, chunk=2, gather=True,
                           on_chunk=lambda lo, hi, fr: seen.append((lo, hi, fr[:, 0, 0, 0].clone())), ring_slots=2).run()
        torch.cuda.synchronize()
        if rank == 0:
            got = sorted(int(v) for lo, hi, t in seen for v in t.tolist())
            ok = ok and got == list(range(23))
        q.put((rank, "ok" if ok else "WRONG DATA"))
        dist.barrier(); dist.destroy_process_group()
    except Exception as e:
        q.put((rank, "ERROR " + repr(e)[:400]))


if __name__ == "__main__":
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    for _ in range(2):
        try:
            print(q.get(timeout=120))
        except Exception as e:
            print("timeout", e)
    for p in ps:
        p.join(timeout=20)
        if p.is_alive(): p.terminate()
