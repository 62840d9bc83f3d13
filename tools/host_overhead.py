#!/usr/bin/env python3
"""Host-side cost of one student Poser.pose() (GPU box): CPU submission time with the queue never full, cProfile of the
Python layer, and the raw C-ABI call loop for comparison."""
import cProfile, ctypes as C, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tha4_amd  # noqa
from tha4_amd import _capi
from tha4_amd.poser.modes import mode_14
from tha4_amd.weights import split_flat_weights
dev = torch.device("cuda:0")
g = os.path.join(ROOT, "tests", "golden")
w = dict(np.load(os.path.join(g, "student_lambda_00_weights.npz")))
io = np.load(os.path.join(g, "student_lambda_00_io.npz"))
p = mode_14.create_poser_from_state_dicts(dev, *split_flat_weights(w), max_batch=4)
image = torch.from_numpy(io["image_f32"]).to(dev)
poses = torch.from_numpy(np.repeat(io["poses"], 64, 0)).to(dev)
N = 200
with torch.no_grad():
    for i in range(50): p.pose(image, poses[i])
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(N): p.pose(image, poses[i])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"pose(): CPU submit {1e6*(t1-t0)/N:.1f} us/frame, until GPU idle {1e6*(t2-t0)/N:.1f} us/frame")
    out = torch.empty(1, 4, 512, 512, device=dev)
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(N): p.pose(image, poses[i], out=out)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"pose(out=): CPU submit {1e6*(t1-t0)/N:.1f} us/frame, until GPU idle {1e6*(t2-t0)/N:.1f} us/frame")
    # raw C ABI
    lib, h = p._lib, p._handle
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ip, op = image.data_ptr(), out.data_ptr()
    pp = [poses[i].data_ptr() for i in range(N)]
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(N): lib.tha4_student_pose(h, ip, 4 * 512 * 512, pp[i], 1, op, None, stream)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"raw tha4_student_pose: CPU submit {1e6*(t1-t0)/N:.1f} us/frame, until GPU idle {1e6*(t2-t0)/N:.1f} us/frame")
    pr = cProfile.Profile(); pr.enable()
    for i in range(N): p.pose(image, poses[i])
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
