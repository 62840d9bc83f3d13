#!/usr/bin/env python3
"""Is the full-model batch-1 stream host-bound?  (GPU box.)  Host time to SUBMIT a short burst of frames into an empty queue
(no back-pressure: a frame is ~360 launches, the burst stays below the queue depth) against the GPU time of the same frames."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tha4_amd  # noqa
from tha4_amd.poser.modes import mode_07
from tha4_amd import synthetic as fo
dev = torch.device("cuda:0")
p = mode_07.create_poser_from_state_dicts(dev, fo.synth_full_weights())
io = np.load(os.path.join(ROOT, "tests/golden/student_lambda_00_io.npz"))
image = torch.from_numpy(io["image_f32"]).to(dev)
poses = torch.from_numpy(io["poses"]).to(dev)
for i in range(5): p.pose(image, poses[i % 8])
torch.cuda.synchronize()
for burst in (1, 2, 4, 8, 16, 60):
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(burst): p.pose(image, poses[i % 8])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r = (1e3 * (t1 - t0) / burst, 1e3 * (t2 - t0) / burst)
        best = r if best is None or r[1] < best[1] else best
    print(f"burst of {burst:3d} frames: host submit {best[0]:.2f} ms/frame, until GPU idle {best[1]:.2f} ms/frame")
print("os.cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads())
