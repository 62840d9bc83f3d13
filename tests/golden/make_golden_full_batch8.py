#!/usr/bin/env python3
"""Golden fixture for the full model (mode_07) at the batch `bench.py --model full --batch 8` runs (BASELINE configs[4]:
8 frames per GPU), from the UNMODIFIED reference:

    python tests/golden/make_golden_full_batch8.py      (build container only)

`full_batch8_io.npz`: `poser.get_posing_outputs(image[8], pose[8])` with 8 DISTINCT images of the SURVEY.md §8d
config-5 recipe (oracle.student_oracle.synthetic_image, seeds 201..208), 8 poses (seed 779), standard synthetic weights
(seed 20260925).  Stored: a stride-7 pixel subset of the distiller's outputs 0,1,2,3,5
(src/tha4/nn/siren/morpher/siren_morpher_protocols_03.py:56-72,102-108) for all eight frames (fp32 run), and of ALL 33
outputs for frame 6 (fp32 and fp64 runs: the fp64 run is the yardstick of the reference's own fp32 scatter).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from make_golden_full_batch import run_both, SEED, TEACHER_OUTPUTS  # noqa: E402  (reference imports inside)

from oracle import full_oracle as fo  # noqa: E402
from oracle.student_oracle import random_poses, synthetic_image  # noqa: E402

SUB7 = slice(3, None, 7)
FRAME = 6


def main():
    seeds = list(range(201, 209))
    images = np.stack([synthetic_image(seed=s) for s in seeds])
    poses = random_poses(8, seed=779)
    ref32, ref64 = run_both(fo.synth_full_weights(SEED), images, poses)
    io = {"image_seeds": np.array(seeds), "poses": poses, "seed": np.int64(SEED), "frame": np.int64(FRAME)}
    for k in TEACHER_OUTPUTS:
        io[f"ref32_sub7_out{k}"] = ref32[k][:, :, SUB7, SUB7]
    for k in range(33):
        io[f"ref32_frame_sub7_out{k}"] = ref32[k][FRAME][:, SUB7, SUB7]
        io[f"ref64_frame_sub7_out{k}"] = ref64[k][FRAME][:, SUB7, SUB7].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "full_batch8_io.npz"), **io)
    noise = {fo.OUTPUT_NAMES[k]: float(np.abs(ref32[k] - ref64[k]).max()) for k in range(33)}
    print("batch-8 fp32-vs-fp64:", {k: f"{v:.2e}" for k, v in noise.items()})


if __name__ == "__main__":
    main()
