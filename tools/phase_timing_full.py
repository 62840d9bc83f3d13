#!/usr/bin/env python3
"""Phase timing of conv_tile_kernel (GPU box): `build` compiles a -DTHA4_PHASE_TIMING variant; the run selects a few
convolutions of the full model by shape, poses one frame each with THA4_DBG_CONV=<index> and prints where a wave's
cycles go (s_memtime stamps at every barrier).  Tuning aid."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "talking-head-anime-4-demo_amd", "csrc")
lib = os.path.join(ROOT, "build_variants", "libtha4_phase.so")
if "build" in sys.argv:
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    from importlib import util as _u
    _sp = _u.spec_from_file_location("_tha4_build", os.path.join(ROOT, "talking-head-anime-4-demo_amd", "_build.py"))
    _b = _u.module_from_spec(_sp)
    _sp.loader.exec_module(_b)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DTHA4_PHASE_TIMING", *_b.DEVICE_FLAGS, *(["-DTHA4_PHASE_WINDOW"] if "--window" in sys.argv else []), "-I", CSRC, "-I",
                    os.path.join(ROOT, "include"), os.path.join(CSRC, "tha4_capi.hip"), "-o", lib], check=True)
    sys.exit(0)

import numpy as np
import torch

os.environ["THA4_HIP_LIB"] = lib
os.environ["THA4_DUMP_SCHEDULE"] = "1"
import tha4_amd  # noqa
from tha4_amd import synthetic
from tha4_amd.poser.modes import mode_07

dev = torch.device("cuda:0")
# capture the schedule dump (stderr of the native library)
tmp = tempfile.TemporaryFile(mode="w+b")
saved = os.dup(2)
os.dup2(tmp.fileno(), 2)
BATCH = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 1
p = mode_07.create_poser_from_state_dicts(dev, synthetic.synth_full_weights(), max_batch=BATCH)
p.get_modules() if hasattr(p, "get_modules") else None
io = np.load(os.path.join(ROOT, "tests/golden/student_lambda_00_io.npz"))
if BATCH == 1:
    image = torch.from_numpy(io["image_f32"]).to(dev)
    poses = torch.from_numpy(io["poses"]).to(dev)
else:
    image = torch.from_numpy(synthetic.random_rgba_images(BATCH, seed=99)).to(dev)
    poses = torch.from_numpy(np.stack([np.resize(io["poses"], (BATCH, 45)) for _ in range(8)])).to(dev)
p.pose(image, poses[0])
torch.cuda.synchronize()
os.dup2(saved, 2)
tmp.seek(0)
sched = [l for l in tmp.read().decode().splitlines() if l.startswith("conv #")]
del os.environ["THA4_DUMP_SCHEDULE"]

targets = ["tile=64x64 cin=256(cb 16) cout=256", "tile=16x16 cin=512(cb 32) cout=512", "tile=256x256 cin=128(cb 8) cout=128",
           "tile=128x128 cin=128(cb 8) cout=128", "tile=32x32 cin=256(cb 16) cout=256", "tile=16x16 cin=256(cb 16) cout=256",
           "tile=128x128 cin=256(cb 16) cout=256", "tile=256x256 cin=64(cb 4) cout=64", "tile=512x512 cin=32(cb 2) cout=32"]
if "--targets" in sys.argv:
    targets = sys.argv[sys.argv.index("--targets") + 1].split(";")
if "--small" in sys.argv:
    targets = [t for t in targets if "16x16" in t or "32x32" in t]
L = p._lib
L.tha4_full_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for tgt in targets:
    line = next((l for l in sched if tgt in l and "kind=0" in l and "mode=0" in l), None)
    if line is None:
        print("no conv matches", tgt)
        continue
    kv = dict(re.findall(r"(\w+)=([\w()x]+)", line))
    idx = int(re.match(r"conv #(\d+)", line).group(1))
    wgs = int(kv["wgs"]) // max(1, int(kv["ksplit"])) * BATCH
    NWV = int(kv.get("nw", "8"))
    os.environ["THA4_DBG_CONV"] = str(idx)
    buf = np.zeros(wgs * NWV * 64, np.int64)
    L.tha4_full_debug_clear(p._handle) if hasattr(L, 'tha4_full_debug_clear') else None
    # clear, run, read
    for _ in range(2):
        p.pose(image, poses[1], image_changed=True)
    assert L.tha4_full_debug_read(p._handle, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    t = buf.reshape(wgs, NWV, 64).astype(np.float64)
    n = int((t[0, 0] > 0).sum())
    first = t[:, :, 0].copy()
    d = np.diff(t[:, :, :n], axis=-1)
    print(f"== {line[:170]}")
    print(f"   stamps per wave: {n}; wave span entry -> last stamp: {(t[:, :, n - 1] - t[:, :, 0]).mean():.0f} cycles; "
          f"first entry -> last stamp over the grid: {t[:, :, n - 1].max() - t[:, :, 0].min():.0f}; entry spread {t[:, :, 0].max() - t[:, :, 0].min():.0f}")
    print("   mean cycles between consecutive stamps: " + " ".join(f"{x:.0f}" for x in d.mean(axis=(0, 1))))
    if "tiled=1" in line:        # conv_tile_kernel: how the workgroups are spread in time (rounds, co-residency)
        ent, end = t[:, 0, 0] - t[:, 0, 0].min(), t[:, 0, n - 1] - t[:, 0, 0].min()
        span = end.max()
        conc = [(int(((ent <= x) & (end > x)).sum())) for x in np.linspace(0.05, 0.95, 10) * span]
        print(f"   kernel span {span:.0f} cycles; workgroups alive at 5 %, 15 %, ... 95 % of it: {conc}; mean workgroup life {np.mean(end - ent):.0f}")
    if "tiled=2" in line:       # conv_small_kernel: waves have different stamp counts (wave 0 runs the epilogue): per-wave rows of workgroup 0 and the spread of entry times
        for wv in (0, 1, 7):
            nn = int((t[0, wv] > 0).sum())
            print(f"   wg 0 wave {wv}: " + " ".join(f"{x:.0f}" for x in np.diff(t[0, wv, :nn])))
        ent = t[:, 0, 0]
        last = np.array([t[w, 0, int((t[w, 0] > 0).sum()) - 1] for w in range(wgs)])
        print(f"   workgroup entry spread: {ent.max() - ent.min():.0f} cycles; first entry -> last epilogue: {last.max() - ent.min():.0f} cycles (s_memtime ticks at 100 MHz)")
os.environ.pop("THA4_DBG_CONV", None)
