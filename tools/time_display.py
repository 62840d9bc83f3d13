#!/usr/bin/env python3
"""Fused display epilogue: stream rate and per-kernel time with / without the RGBA8 output (GPU box; tuning aid)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tha4_amd  # noqa
from tha4_amd.poser.modes import mode_14
from tha4_amd.weights import split_flat_weights
from tha4_amd import image_io
g = os.path.join(ROOT, "tests", "golden")
w = dict(np.load(os.path.join(g, "student_lambda_00_weights.npz"))); io = np.load(os.path.join(g, "student_lambda_00_io.npz"))
f, b = split_flat_weights(w)
dev = torch.device("cuda:0")
p = mode_14.create_poser_from_state_dicts(dev, f, b, max_batch=4)
img = torch.from_numpy(io["image_f32"]).to(dev)
poses = torch.from_numpy(np.repeat(io["poses"], 64, 0)).to(dev)
out_f = torch.empty((1, 4, 512, 512), device=dev)
out_u = torch.empty((1, 512, 512, 4), dtype=torch.uint8, device=dev)
def rate(fn, n=1000):
    for i in range(50): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return n / (time.perf_counter() - t0)
print("pose(out=)                      %8.1f fps" % rate(lambda i: p.pose(img, poses[i % 512], out=out_f)))
print("pose_display_rgba8(out=)        %8.1f fps" % rate(lambda i: p.pose_display_rgba8(img, poses[i % 512], out=out_u)))
print("pose_display_rgba8(out=, bg)    %8.1f fps" % rate(lambda i: p.pose_display_rgba8(img, poses[i % 512], background_rgb=(0, 1, 0), out=out_u)))
print("pose_display_rgba8(want_frame)  %8.1f fps" % rate(lambda i: p.pose_display_rgba8(img, poses[i % 512], out=out_u, want_frame=True)))
print("pose + to_display_rgba8         %8.1f fps" % rate(lambda i: image_io.to_display_rgba8(p.pose(img, poses[i % 512], out=out_f))))
p.set_timing(True)
for name, fn in (("pose", lambda i: p.pose(img, poses[i], out=out_f)), ("rgba8", lambda i: p.pose_display_rgba8(img, poses[i], out=out_u))):
    acc = np.zeros(5)
    for i in range(100):
        fn(i)
        for k in range(5): acc[k] += p.last_kernel_ms(k)
    print(name, "kernel ms (posebias, face, level0/front, level1, level2):", np.round(acc / 100, 4))
