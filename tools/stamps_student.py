#!/usr/bin/env python3
"""Read the in-kernel time stamps of a -DTHA4_STAMPS tuning build (GPU box): where does a wave of front16r_kernel / level1_16r_kernel spend its launch?

  THA4_HIP_LIB=build_variants/libtha4_stamps.so python tools/stamps_student.py

Slots: 0 = level-0 workgroup 0 wave 0, 1 = level-0 workgroup 131 wave 3, 2 = face workgroup 0 wave 0, 3 = level-1 workgroup 0 wave 0.  Stamps are shader
clocks (s_memtime); printed as microseconds at 2.4 GHz relative to the slot's entry stamp, median over the frames."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tha4_amd  # noqa: E402,F401
from tha4_amd.poser.modes import mode_14  # noqa: E402
from tha4_amd.weights import split_flat_weights  # noqa: E402

L0 = {0: "entry", 1: "tables in LDS (pose fold, biases, wx, wy)", 2: "ring requested + barrier", 3: "first layer done", 4: "layer A GEMM done (24 chunks)", 5: "sine A done",
      6: "layer B GEMM done (12 chunks)", 7: "sine B done", 8: "z GEMM done (6 chunks)", 9: "z stores issued"}
for i in range(4):
    L0[16 + 3 * i], L0[17 + 3 * i], L0[18 + 3 * i] = f"chunk {20 + i}: at top", f"chunk {20 + i}: landed + barrier passed + next requested", f"chunk {20 + i}: MFMAs issued"
FACE = {0: "entry", 2: "prologue done", 3: "first layer done", 12: "end"}
FACE.update({4 + l: f"layer {l + 1} done" for l in range(7)})
L1 = {0: "entry", 2: "prologue done", 3: "first layer (taps) done", 4: "layer A GEMM done", 5: "sine A done", 6: "layer B GEMM done", 7: "sine B done", 8: "z GEMM done", 9: "z stores issued"}


def main():
    g = os.path.join(ROOT, "tests", "golden")
    w = dict(np.load(os.path.join(g, "student_lambda_00_weights.npz")))
    io = np.load(os.path.join(g, "student_lambda_00_io.npz"))
    dev = torch.device("cuda:0")
    poser = mode_14.create_poser_from_state_dicts(dev, *split_flat_weights(w))
    image = torch.from_numpy(io["image_f32"]).to(dev)
    poses = torch.from_numpy(io["poses"]).to(dev)
    rows, spans = [], []
    host = torch.empty(800, dtype=torch.float32)
    init = torch.zeros(800, dtype=torch.float32).numpy().view(np.uint64)
    init[320:340:2] = np.uint64(2 ** 63)                         # span slots: min fields start high, max fields at 0
    poser.pose(image, poses[0])                                  # (the handle is created lazily)
    torch.cuda.synchronize()
    for i in range(40):
        # (the stamps live in the handle's pose-bias workspace: reset the min / max fields through the debug read's device pointer - a tuning-build-only backdoor)
        poser._lib.tha4_student_debug_write(poser._handle, C.c_void_p(init.ctypes.data))
        poser.pose(image, poses[i % poses.shape[0]])
        torch.cuda.synchronize()
        rc = poser._lib.tha4_student_debug_read(poser._handle, 2, 0, C.c_void_p(host.data_ptr()))
        assert rc == 0, "not a -DTHA4_STAMPS build?"
        if i >= 8:
            rows.append(host.numpy().view(np.uint64).copy())
            spans.append(host.numpy().view(np.uint64)[320:340].astype(np.int64).copy())
    a = np.stack(rows).astype(np.int64)                      # [frames, 400]
    for slot, names, title in ((0, L0, "level 0, workgroup 0 wave 0"), (1, L0, "level 0, workgroup 131 wave 3"), (2, FACE, "face, workgroup 0 wave 0"), (3, L1, "level 1, workgroup 0 wave 0")):
        s = a[:, slot * 64:(slot + 1) * 64]
        print(f"== {title}")
        prev = 0.0
        for k in sorted(names):
            t = float(np.median(s[:, k] - s[:, 0])) / 2400.0
            print(f"  {t:8.2f} us  (+{t - prev:6.2f})  {names[k]}")
            prev = t


    sp = np.stack(spans)
    print("== workgroup entry / exit spans in the 100-MHz wall clock, us relative to the front kernel's first workgroup entry (median over frames)")
    t0 = sp[:, 0:1]
    names = ["level 0 wgs", "face wgs", "level 1 wgs", "level 2 wgs"]
    for k in range(4):
        v = np.median((sp[:, 4 * k:4 * k + 4] - t0) / 100.0, axis=0)
        print(f"  {names[k]:12s} first entry {v[0]:7.2f}  last entry {v[1]:7.2f}  first exit {v[2]:7.2f}  last exit {v[3]:7.2f}")


    v = np.median((sp[:, 16:20] - t0) / 100.0, axis=0)
    print(f"  prologue done (tables in LDS, barrier passed): level 0 wgs earliest {v[0]:7.2f} latest {v[1]:7.2f}   face wgs earliest {v[2]:7.2f} latest {v[3]:7.2f}")


if __name__ == "__main__":
    main()
