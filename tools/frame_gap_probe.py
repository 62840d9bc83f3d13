#!/usr/bin/env python3
"""Where does the ~4.4 us gap in front of the first kernel of every student frame come from?  Three loops of 300 frames each,
separated by a marker kernel (display_rgba8 of a tiny frame): (a) Poser.pose(), (b) Poser.pose(out=preallocated), (c) the raw
C ABI call with fixed pointers.  Run under rocprofv3 --kernel-trace and read the gaps with tools/trace_gaps.py-style logic."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tha4_amd  # noqa
from tha4_amd import image_io
from tha4_amd.poser.modes import mode_14
from tha4_amd.weights import split_flat_weights
dev = torch.device("cuda:0")
g = os.path.join(ROOT, "tests", "golden")
w = dict(np.load(os.path.join(g, "student_lambda_00_weights.npz")))
io = np.load(os.path.join(g, "student_lambda_00_io.npz"))
p = mode_14.create_poser_from_state_dicts(dev, *split_flat_weights(w), max_batch=4)
image = torch.from_numpy(io["image_f32"]).to(dev)
poses = torch.from_numpy(np.repeat(io["poses"], 64, 0)).to(dev)
marker = torch.zeros(1, 4, 8, 8, device=dev)
N = 300
with torch.no_grad():
    for i in range(50): p.pose(image, poses[i])
    torch.cuda.synchronize(); image_io.to_display_rgba8(marker); torch.cuda.synchronize()
    for i in range(N): p.pose(image, poses[i])
    torch.cuda.synchronize(); image_io.to_display_rgba8(marker); torch.cuda.synchronize()
    out = torch.empty(1, 4, 512, 512, device=dev)
    for i in range(N): p.pose(image, poses[i], out=out)
    torch.cuda.synchronize(); image_io.to_display_rgba8(marker); torch.cuda.synchronize()
    lib, h = p._lib, p._handle
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ip, op = image.data_ptr(), out.data_ptr()
    pp = [poses[i].data_ptr() for i in range(N)]
    for i in range(N): lib.tha4_student_pose(h, ip, 4 * 512 * 512, pp[i], 1, op, None, stream)
    torch.cuda.synchronize(); image_io.to_display_rgba8(marker); torch.cuda.synchronize()
