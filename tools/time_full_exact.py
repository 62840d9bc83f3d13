#!/usr/bin/env python3
"""Frames/s of the full model on the exact-fp32 plan (THA4_FULL_EXACT_FP32) next to the default plan (GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tha4_amd  # noqa
from tha4_amd.poser.modes import mode_07
from tha4_amd import synthetic as fo
dev = torch.device("cuda:0")
w = fo.synth_full_weights()
io = np.load(os.path.join(ROOT, "tests/golden/student_lambda_00_io.npz"))
image = torch.from_numpy(io["image_f32"]).to(dev)
poses = torch.from_numpy(io["poses"]).to(dev)
for exact in (False, True):
    p = mode_07.create_poser_from_state_dicts(dev, w, exact_fp32=exact)
    for i in range(3): p.pose(image, poses[i % 8])
    torch.cuda.synchronize()
    for name, changed, n in (("steady", False, 30), ("cold", True, 20)):
        t0 = time.perf_counter()
        for i in range(n): p.pose(image, poses[i % 8], image_changed=changed)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{'exact-fp32 plan' if exact else 'default plan   '} {name}: {n / dt:.2f} frames/s  {1e3 * dt / n:.2f} ms/frame", flush=True)
    p.free()
