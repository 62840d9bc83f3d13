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
BATCH = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 1
if BATCH > 1:
    # configs[4], one GPU's share: BATCH distinct random images + poses per call (always cold), the launch plan of a max_batch = BATCH handle
    p = mode_07.create_poser_from_state_dicts(dev, w, max_batch=BATCH)
    imgs = [torch.from_numpy(fo.random_rgba_images(BATCH, seed=99 + j)).to(dev) for j in range(2)]
    g = torch.Generator().manual_seed(77)
    lo = torch.tensor([0.0] * 37 + [-1.0] * 7 + [0.0]); hi = torch.ones(45)
    poses = (lo + (hi - lo) * torch.rand(16, BATCH, 45, generator=g)).to(dev)
    for i in range(3): p.pose(imgs[i % 2], poses[i])
    torch.cuda.synchronize()
    n = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 10
    t0 = time.perf_counter()
    for i in range(n): p.pose(imgs[i % 2], poses[i % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"full model batch {BATCH}: {n * BATCH / dt:.2f} fps  {1e3 * dt / n:.2f} ms/step ({n} steps + 3 warm-up)")
    sys.exit(0)
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
