#!/usr/bin/env python3
"""Pose-sweep fixture for the student (mode_14), SURVEY.md §8d config 2 ("parity on the first 64 frames"), from the
UNMODIFIED reference:

    python tests/golden/make_golden_sweep.py [lambda_00|lambda_01] [n_pinned]      (build container only)

`student_<name>_sweep.npz`: poses[64,45] = the first 64 poses of the config-2 stream (oracle.student_oracle.random_poses,
seed 1234: the first 8 are the poses of student_<name>_io.npz) and, for the first `n_pinned` of them (default: ALL 64, both
characters - round 4; round 3 pinned 16 of lambda_01), a stride-8 pixel subset (offset 3) of the posed frame (output 0) of
`mode_14.create_poser(...).pose(image, pose)`, fp32, 8 threads.  The device test poses all 64 frames, compares the pinned
ones with these subsets and every frame, full size, with the oracle evaluated on the box.
Round 4 additions: `edge_poses[4,45]` (all zeros, every parameter at its lower / upper limit, alternating limits - the poses of
tests/test_student_gpu.py::test_edge_poses) with `ref32_sub8_edge_out0` from the unmodified reference; `torch_version` and
`aten_axis{128,256,512}` = the fp32 `affine_grid` axes of the torch build that made the fixture (the kernels take the LOCAL
build's table so that they track "the reference on this machine": tests/test_oracle_golden.py warns when the two differ).
"""
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

from oracle.student_oracle import POSE_HI, POSE_LO, random_poses  # noqa: E402

SUB8 = slice(3, None, 8)
N_POSES = 64


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "lambda_00"
    assert name in ("lambda_00", "lambda_01"), name
    n_pinned = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    cm = os.path.join(REF, "data/character_models", name)
    poser = create_poser(torch.device("cpu"), module_file_names={"face_morpher": os.path.join(cm, "face_morpher.pt"),
                                                                 "body_morpher": os.path.join(cm, "body_morpher.pt")})
    image = extract_pytorch_image_from_PIL_image(PIL.Image.open(os.path.join(cm, "character.png")))
    poses = random_poses(N_POSES, seed=1234)
    torch.set_num_threads(8)
    subs = []
    with torch.no_grad():
        for i in range(n_pinned):
            subs.append(poser.pose(image, torch.from_numpy(poses[i]))[0].numpy()[:, SUB8, SUB8].copy())
    edge = np.stack([np.zeros(45, np.float32), POSE_LO, POSE_HI,
                     np.where(np.arange(45) % 2 == 0, POSE_LO, POSE_HI).astype(np.float32)])
    edge_subs = []
    with torch.no_grad():
        for i in range(edge.shape[0]):
            edge_subs.append(poser.pose(image, torch.from_numpy(edge[i]))[0].numpy()[:, SUB8, SUB8].copy())
    ident = torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]])
    axes = {f"aten_axis{sz}": torch.nn.functional.affine_grid(ident, [1, 1, sz, sz], align_corners=False)[0, 0, :, 0].numpy().copy()
            for sz in (128, 256, 512)}
    np.savez_compressed(os.path.join(HERE, f"student_{name}_sweep.npz"), poses=poses, ref32_sub8_out0=np.stack(subs),
                        edge_poses=edge, ref32_sub8_edge_out0=np.stack(edge_subs), torch_version=np.array(torch.__version__), **axes)
    print(name, "pinned", n_pinned, "poses +", len(edge_subs), "edge poses; subset shape", subs[0].shape, "; torch", torch.__version__)


if __name__ == "__main__":
    main()
