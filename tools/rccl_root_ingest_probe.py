#!/usr/bin/env python3
"""What the ROOT of the frame gather pays on its own device (round-4 review, task 2c).  A 1-GPU lease cannot measure xGMI, but it can
measure rank 0's side of `bench.py --gpus 8`: while rank 0 poses its own frames, the frames of the 7 other ranks land in its HBM - at
8 x 8000 frames/s of fp32 frames that is 7 x 8000 x 4.19 MB = 235 GB/s of incoming writes (59 GB/s as RGBA8), plus the consumer's pass
over them.  Here the arriving frames are device-to-device copies on a side stream (the same destination writes RCCL's receive kernels
perform, from a source in local HBM instead of a peer's: an UPPER bound on the HBM-side interference, no statement about the links),
the own block goes through `dist.gather` in a one-rank RCCL group (`force_collective`: the real call path), and the student stream is
timed with and without that traffic.

    python tools/rccl_root_ingest_probe.py [--frames 2000] [--world 8]        (GPU box)

Prints frames/s of rank 0's own stream alone, with the emulated ingest of (world - 1) peers at its own rate (fp32 and RGBA8), and the
achieved ingest rate."""
import argparse
import os
import socket
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=32)
    args = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import tha4_amd  # noqa: F401
    from tha4_amd.poser.modes import mode_14
    from tha4_amd.sharding import FrameShardedStream
    from tha4_amd.weights import split_flat_weights
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = os.path.join(ROOT, "tests", "golden")
    w = dict(np.load(os.path.join(g, "student_lambda_00_weights.npz")))
    io = np.load(os.path.join(g, "student_lambda_00_io.npz"))
    poser = mode_14.create_poser_from_state_dicts(dev, *split_flat_weights(w), max_batch=4)
    image = torch.from_numpy(io["image_f32"]).to(dev)
    poses = torch.from_numpy(np.resize(io["poses"], (args.frames, 45))).to(dev)
    peers = args.world - 1
    C = args.chunk

    def run(mode):
        """mode: 'alone' | 'fp32' | 'rgba8'"""
        rgba8 = mode == "rgba8"
        shape, dtype = ((512, 512, 4), torch.uint8) if rgba8 else ((4, 512, 512), torch.float32)
        step = poser.pose_display_rgba8 if rgba8 else poser.pose
        src = torch.empty((peers * C,) + shape, dtype=dtype, device=dev)                 # what the peers would send per round
        ring = [torch.empty((peers * C,) + shape, dtype=dtype, device=dev) for _ in range(3)]
        side = torch.cuda.Stream(device=dev)
        rounds = [0]

        def frame_fn(lo, hi):
            blk = torch.empty((hi - lo,) + shape, dtype=dtype, device=dev)
            for i in range(lo, hi):
                step(image, poses[i], out=blk[i - lo:i - lo + 1])
            if mode != "alone":                    # the peers' frames of this round arrive while the next round is posed
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    ring[rounds[0] % 3].copy_(src, non_blocking=True)
                rounds[0] += 1
            return blk

        seen = [0]
        st = FrameShardedStream(frame_fn, args.frames, shape, dtype, dev, chunk=C, gather=True, force_collective=True,
                                on_chunk=lambda lo, hi, fr: seen.__setitem__(0, seen[0] + hi - lo), ring_slots=3)
        with torch.no_grad():
            st.run()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            st.run()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
        fps = args.frames / dt
        ingest = 0.0 if mode == "alone" else peers * args.frames * src[0].numel() * src.element_size() / dt / 1e9
        return fps, ingest

    base, _ = run("alone")
    print(f"rank 0 alone (own frames through a one-rank RCCL gather, chunk {C}): {base:.1f} frames/s")
    for mode in ("fp32", "rgba8"):
        fps, ingest = run(mode)
        print(f"rank 0 with the emulated ingest of {peers} peers at its own rate, {mode} frames: {fps:.1f} frames/s ({fps / base * 100:.1f} % of alone), "
              f"{ingest:.1f} GB/s of arriving frames written to its HBM")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
