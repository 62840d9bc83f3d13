#!/usr/bin/env python3
"""Generate the golden fixtures for the student (mode_14) path by running the UNMODIFIED
reference (imported read-only from /root/reference/src) on CPU.

Run in the build container only (the GPU box has no /root/reference):

    python tests/golden/make_golden.py [lambda_00|lambda_01]     (default lambda_00)

Both shipped students are covered; `student_<name>_*.npz` have the same layout.

Writes (all under tests/golden/):
  student_lambda_00_weights.npz   the two shipped state_dicts, flattened to fp32 arrays
                                  (keys 'face.<state_dict key>' / 'body.<state_dict key>',
                                  1x1 conv kernels squeezed to [out,in])
  student_lambda_00_io.npz        image_rgba8 (decoded character.png), image_f32 (the tensor the
                                  reference's extract_pytorch_image_from_PIL_image returns),
                                  poses[8,45] (seed 1234), and reference outputs:
                                    ref32_full_out0        [1 pose][4,512,512]   fp32, thread count 8
                                    ref32_sub_out{0..5}    [3 poses] stride-3 pixel subset, fp32
                                    ref64_sub_out{0..5}    [3 poses] same subset, reference modules .double(), rounded to fp32 for storage
  student_lambda_00_noise.json    the reference's own noise floor (1 vs N threads, fp32 vs fp64)

Data licence: the lambda_00 / lambda_01 character images and student weights are (c) Pramook Khungurn,
CC BY-NC 4.0 (reference README.md:273-274, data/images/README.md).
"""
import json
import os
import sys

import numpy as np
import PIL.Image
import torch

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tha4.poser.modes.mode_14 import create_poser  # noqa: E402  (reference, unmodified)
from tha4.shion.base.image_util import extract_pytorch_image_from_PIL_image  # noqa: E402

from oracle.student_oracle import random_poses, state_dicts_to_numpy  # noqa: E402

SUB = slice(1, None, 3)      # stride-3 pixel subset: hits both parities of the 2x upsample taps
N_POSES = 8
N_REF = 3


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "lambda_00"
    assert name in ("lambda_00", "lambda_01"), name
    cm = os.path.join(REF, "data/character_models", name)
    files = {"face_morpher": os.path.join(cm, "face_morpher.pt"),
             "body_morpher": os.path.join(cm, "body_morpher.pt")}
    poser = create_poser(torch.device("cpu"), module_file_names=dict(files))
    pil = PIL.Image.open(os.path.join(cm, "character.png"))
    image = extract_pytorch_image_from_PIL_image(pil)
    rgba8 = np.asarray(pil.convert("RGBA"), dtype=np.uint8)
    poses = random_poses(N_POSES, seed=1234)

    mods = poser.get_modules()
    w = state_dicts_to_numpy(mods["face_morpher"].state_dict(), mods["body_morpher"].state_dict())
    np.savez(os.path.join(HERE, f"student_{name}_weights.npz"), **w)

    io = {"image_rgba8": rgba8, "image_f32": image.numpy(), "poses": poses}
    torch.set_num_threads(8)
    ref32 = []
    with torch.no_grad():
        for i in range(N_REF):
            ref32.append([o[0].numpy().copy() for o in poser.get_posing_outputs(image, torch.from_numpy(poses[i]))])
    io["ref32_full_out0"] = ref32[0][0][None]
    for k in range(6):
        io[f"ref32_sub_out{k}"] = np.stack([r[k][:, SUB, SUB] for r in ref32])

    # noise floor: single-thread fp32
    torch.set_num_threads(1)
    with torch.no_grad():
        ref32_1t = [[o[0].numpy().copy() for o in poser.get_posing_outputs(image, torch.from_numpy(poses[i]))]
                    for i in range(2)]
    torch.set_num_threads(8)

    # fp64 run of the reference modules
    for m in mods.values():
        m.double()
    poser.dtype = torch.float64
    ref64 = []
    with torch.no_grad():
        for i in range(N_REF):
            ref64.append([o[0].numpy().copy()
                          for o in poser.get_posing_outputs(image.double(), torch.from_numpy(poses[i]).double())])
    for k in range(6):
        io[f"ref64_sub_out{k}"] = np.stack([r[k][:, SUB, SUB] for r in ref64]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, f"student_{name}_io.npz"), **io)

    names = ["blended", "alpha", "color_change", "warped", "grid_change", "face"]
    noise = {"torch": torch.__version__, "threads": 8, "poses": N_REF,
             "fp32_8t_vs_1t_maxabs": {names[k]: float(max(np.abs(ref32[i][k] - ref32_1t[i][k]).max() for i in range(2)))
                                      for k in range(6)},
             "fp32_vs_fp64_maxabs": {names[k]: float(max(np.abs(ref32[i][k] - ref64[i][k]).max() for i in range(N_REF)))
                                     for k in range(6)}}
    with open(os.path.join(HERE, f"student_{name}_noise.json"), "w") as f:
        json.dump(noise, f, indent=1)
    print(json.dumps(noise, indent=1))


if __name__ == "__main__":
    main()
