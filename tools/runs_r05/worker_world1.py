"""Smoke of tests/test_sharding_nccl_gpu.py::_worker at world 1 (GPU box): the code path 1-GPU leases skip - parses, runs, reports a rate."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    import torch.multiprocessing as mp
    import test_sharding_nccl_gpu as T
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = ctx.Process(target=T._worker, args=(0, 1, port, 6, 4, q))
    p.start()
    print("result", q.get(timeout=240))
    p.join(timeout=60)
    print("exit", p.exitcode)
