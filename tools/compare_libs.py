#!/usr/bin/env python3
"""Bitwise comparison of two builds of libtha4_hip.so over the first N poses of the config-2 stream (GPU box; tuning aid).

  python tools/compare_libs.py default build_variants/libtha4_wait0.so [n_poses]

Used to validate a shipped build against the same source compiled with `-mllvm -amdgpu-waitcnt-forcezero=1` (every memory
wait forced to zero): the arithmetic is identical, so any differing byte is a memory-ordering / hazard fault in one of them."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import os, sys, numpy as np, torch, hashlib
sys.path.insert(0, %r)
import tha4_amd
from tha4_amd.poser.modes import mode_14
from tha4_amd.weights import split_flat_weights
from oracle.student_oracle import random_poses
g = os.path.join(%r, "tests", "golden")
n = int(sys.argv[2])
out = []
for ch in ("lambda_00", "lambda_01"):
    w = dict(np.load(os.path.join(g, f"student_{ch}_weights.npz"))); io = np.load(os.path.join(g, f"student_{ch}_io.npz"))
    f, b = split_flat_weights(w)
    p = mode_14.create_poser_from_state_dicts(torch.device("cuda:0"), f, b, max_batch=8)
    img = torch.from_numpy(io["image_f32"]).cuda()
    poses = torch.from_numpy(random_poses(n, seed=1234)).cuda()
    for i0 in range(0, n, 8):
        outs = p.get_posing_outputs(img, poses[i0:i0 + 8])
        for k in range(outs[0].shape[0]):
            out.append(hashlib.sha1(b"".join(o[k].cpu().numpy().tobytes() for o in outs)).hexdigest())
        one = p.get_posing_outputs(img, poses[i0])
        out.append(hashlib.sha1(b"".join(o[0].cpu().numpy().tobytes() for o in one)).hexdigest())
open(sys.argv[1], "w").write("\n".join(out))
""" % (ROOT, ROOT)


def run(lib, n, tag):
    env = dict(os.environ)
    env.pop("THA4_HIP_LIB", None)
    if lib != "default":
        env["THA4_HIP_LIB"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
    out = f"/tmp/compare_libs_{tag}.txt"
    subprocess.run([sys.executable, "-c", CODE, out, str(n)], check=True, env=env)
    return open(out).read().split("\n")


def main():
    a, b = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    ha, hb = run(a, n, "a"), run(b, n, "b")
    diff = sum(x != y for x, y in zip(ha, hb))
    print(f"{a} vs {b}: {len(ha)} evaluations (both characters, {n} poses in batches of 8 + single frames, all six outputs hashed); differing: {diff}")
    sys.exit(1 if diff else 0)


if __name__ == "__main__":
    main()
