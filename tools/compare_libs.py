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


CODE_FULL = r"""
import os, sys, numpy as np, torch, hashlib
sys.path.insert(0, %r)
import tha4_amd
from tha4_amd import synthetic
from tha4_amd.poser.modes import mode_07
from oracle.student_oracle import random_poses, synthetic_image
g = os.path.join(%r, "tests", "golden")
n = int(sys.argv[2])
io = np.load(os.path.join(g, "student_lambda_00_io.npz"))
w = synthetic.synth_full_weights()
out = []
dev = torch.device("cuda:0")
for mb in (1, 4, 8):
    p = mode_07.create_poser_from_state_dicts(dev, w, max_batch=mb)
    imgs = torch.from_numpy(np.stack([io["image_f32"]] + [synthetic_image(seed=300 + i) for i in range(mb - 1)])).to(dev) if mb > 1 else torch.from_numpy(io["image_f32"]).to(dev)
    poses = torch.from_numpy(random_poses(n * mb, seed=4321)).to(dev)
    for i in range(n):
        outs = p.get_posing_outputs(imgs, poses[i * mb:(i + 1) * mb] if mb > 1 else poses[i], image_changed=(i %% 2 == 0))
        out.append(hashlib.sha1(b"".join(o.cpu().numpy().tobytes() for o in outs)).hexdigest())
    p.free()
open(sys.argv[1], "w").write("\n".join(out))
""" % (ROOT, ROOT)


def run(lib, n, tag):
    env = dict(os.environ)
    env.pop("THA4_HIP_LIB", None)
    if lib != "default":
        env["THA4_HIP_LIB"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
    out = f"/tmp/compare_libs_{tag}.txt"
    subprocess.run([sys.executable, "-c", CODE_FULL if "--full" in sys.argv else CODE, out, str(n)], check=True, env=env)
    return open(out).read().split("\n")


def main():
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    a, b = args[0], args[1]
    n = int(args[2]) if len(args) > 2 else 64
    ha, hb = run(a, n, "a"), run(b, n, "b")
    diff = sum(x != y for x, y in zip(ha, hb))
    what = ("full model: handles for 1 / 4 / 8 frames, all 33 outputs hashed, steady and cold calls" if "--full" in sys.argv else
            f"both characters, {n} poses in batches of 8 + single frames, all six outputs hashed")
    print(f"{a} vs {b}: {len(ha)} evaluations ({what}); differing: {diff}")
    sys.exit(1 if diff else 0)


if __name__ == "__main__":
    main()
