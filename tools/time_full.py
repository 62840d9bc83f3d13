#!/usr/bin/env python3
"""Quick timing of the full-model path (GPU box): frames/s steady (decomposer cached) and cold."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tha4_amd  # noqa
from tha4_amd.poser.modes import mode_07
from tha4_amd import synthetic as fo
dev = torch.device("cuda:0")
w = fo.synth_full_weights()
p = mode_07.create_poser_from_state_dicts(dev, w)
io = np.load(os.path.join(ROOT, "tests/golden/student_lambda_00_io.npz"))
image = torch.from_numpy(io["image_f32"]).to(dev)
poses = torch.from_numpy(io["poses"]).to(dev)
for i in range(3): p.pose(image, poses[i % 8])
torch.cuda.synchronize()
NFRAMES = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 20
MODES = {"steady": [("steady", False)], "cold": [("cold", True)], "both": [("steady", False), ("cold", True)]}[
    sys.argv[sys.argv.index("--mode") + 1] if "--mode" in sys.argv else "both"]
for name, changed in MODES:
    n = NFRAMES
    t0 = time.perf_counter()
    for i in range(n): p.pose(image, poses[i % 8], image_changed=changed)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"full model {name}: {n/dt:.2f} fps  {1e3*dt/n:.2f} ms/frame")
