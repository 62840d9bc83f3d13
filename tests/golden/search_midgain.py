#!/usr/bin/env python3
"""Search of the MID-GAIN parameter set of the full model (round-4 review, task 6): per-row gains of the two U-Nets' last
convolution - `synth_full_weights(head_gains=(direct, grid, alpha))` - such that the U-Net outputs are O(0.2-0.3), alpha spans
most of (0, 1) and the UNMODIFIED reference still agrees with itself (fp32 vs fp64 run) to <= 2e-4 on the posed frame.

    python tests/golden/search_midgain.py [kd,kg,ka ...]      (build container only: imports /root/reference)

Prints, per candidate, the ranges of the upscaler's / body morpher's direct, grid and alpha outputs and the reference's own
fp32-vs-fp64 distance per U-Net output.  The triple that tests/golden/make_golden_full_midgain.py uses is recorded there.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from make_golden_full_batch import run_both  # noqa: E402  (reference imports)

from oracle import full_oracle as fo  # noqa: E402
from oracle.student_oracle import random_poses  # noqa: E402

SEED = 20260925


def evaluate(gains, image, poses):
    w = fo.synth_full_weights(SEED, head_gains=gains)
    ref32, ref64 = run_both(w, image, poses)
    names = fo.OUTPUT_NAMES
    row = {}
    for k in range(11):
        row[names[k]] = (float(ref32[k].min()), float(ref32[k].max()), float(np.abs(ref32[k] - ref64[k]).max()))
    return row


def main():
    cands = [tuple(float(x) for x in a.split(",")) for a in sys.argv[1:]] or [(4.0, 2.0, 40.0), (4.0, 4.0, 40.0), (5.0, 3.0, 50.0)]
    image = np.load(os.path.join(HERE, "student_lambda_00_io.npz"))["image_f32"][None]
    poses = random_poses(1, seed=4321)
    for g in cands:
        row = evaluate(g, image, poses)
        print(f"head_gains = {g}")
        for k, (lo, hi, d) in row.items():
            print(f"   {k:18s} range [{lo:+.4f}, {hi:+.4f}]   reference fp32 vs fp64 {d:.2e}")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
