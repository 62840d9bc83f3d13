#!/usr/bin/env python3
"""Does replaying a frame of the full model as ONE hipGraph shorten it?  (GPU box; tuning probe.)  The frame is ~324 dependent launches that the host
already submits ahead of the device; a graph removes the host's per-launch work and lets the runtime pre-build the packets - whether the DEVICE-side
gap between dependent kernels shrinks is what this measures.  tha4_full_pose does no allocation / synchronisation, so it can be captured as it is."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tha4_amd  # noqa
from tha4_amd import synthetic as fo
from tha4_amd.poser.modes import mode_07

dev = torch.device("cuda:0")
w = fo.synth_full_weights()
for B in (1, 8):
    p = mode_07.create_poser_from_state_dicts(dev, w, max_batch=B)
    if B == 1:
        io = np.load(os.path.join(ROOT, "tests/golden/student_lambda_00_io.npz"))
        image = torch.from_numpy(io["image_f32"]).to(dev)
        pose = torch.from_numpy(io["poses"][1]).to(dev)
    else:
        image = torch.from_numpy(fo.random_rgba_images(B, seed=99)).to(dev)
        pose = torch.rand(B, 45, device=dev)
    for _ in range(3):
        ref = p.pose(image, pose)
    torch.cuda.synchronize()
    n = 60 if B == 1 else 10
    t0 = time.perf_counter()
    for _ in range(n):
        p.pose(image, pose)
    torch.cuda.synchronize()
    t_stream = (time.perf_counter() - t0) / n
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                p.pose(image, pose)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = p.pose(image, pose)
        g.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / n
        print(f"batch {B}: stream launches {B / t_stream:.2f} frames/s ({1e3 * t_stream:.3f} ms/step)   one graph per step {B / t_graph:.2f} frames/s ({1e3 * t_graph:.3f} ms/step)   "
              f"graph output == stream output: {same}", flush=True)
    except Exception as e:  # noqa
        print(f"batch {B}: stream launches {B / t_stream:.2f} frames/s; graph capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
    del p
