#!/usr/bin/env python3
"""GPU box: pose one frame three times with level 2 taken from each given code object (THA4_L2_CODE_OBJECT hook of the C ABI) and report
max |posed frame - oracle| and whether the three evaluations are bitwise equal.  Needs a library built with -DTHA4_L2_HOOK
(THA4_HIP_LIB=build_variants/libtha4_hook.so; `THA4_SWEEP_VARIANTS=hook python tools/sweep.py build`).   python tools/hunt/check_co.py dir_or_files..."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNEL = "_ZN4tha42v217level2_16p_kernelILi8ELi32ELi2EEEvNS_10StudentDevE"
CODE = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import tha4_amd
from tha4_amd.poser.modes import mode_14
from tha4_amd.weights import split_flat_weights
g = os.path.join(%r, "tests", "golden")
w = dict(np.load(os.path.join(g, "student_lambda_00_weights.npz"))); io = np.load(os.path.join(g, "student_lambda_00_io.npz"))
f, b = split_flat_weights(w)
p = mode_14.create_poser_from_state_dicts(torch.device("cuda:0"), f, b)
img = torch.from_numpy(io["image_f32"]).cuda()
ref = torch.from_numpy(io["ref32_full_out0"]).cuda()
worst, same = 0.0, True
first = None
for rep in range(10):
    for i in range(1):
        out = p.pose(img, torch.from_numpy(io["poses"][i]).cuda())
        worst = max(worst, float((out - ref).abs().max()))
        if first is None: first = out.clone()
        else: same = same and bool(torch.equal(out, first))
print("RESULT %%.3e %%s" %% (worst, same))
""" % (ROOT, ROOT)


def main():
    files = []
    for a in sys.argv[1:]:
        files += sorted(glob.glob(os.path.join(a, "*.co"))) if os.path.isdir(a) else [a]
    for f in files:
        env = dict(os.environ, THA4_L2_CODE_OBJECT=os.path.abspath(f), THA4_L2_KERNEL=KERNEL, THA4_L2_THREADS="512", THA4_L2_PX="1024")
        r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print(f"{os.path.basename(f):16s}", line[0] if line else "FAILED " + r.stderr[-300:].replace("\n", " | "), flush=True)


if __name__ == "__main__":
    main()
