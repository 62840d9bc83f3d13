#!/usr/bin/env python3
"""Golden fixtures for the FULL model (mode_07) from the UNMODIFIED reference, using the
deterministic synthetic weights of oracle.full_oracle.synth_full_weights (the reference checkout
ships no full-model weights: data/tha4/placeholder.txt).

    python tests/golden/make_golden_full.py      (build container only)

Steps: (1) build the five reference modules with the reference's own factories/arguments
(mode_07.py:137-269) but WITHOUT torch_load; (2) load_state_dict(strict=True) the synthetic
state_dicts - this pins the key/shape layout; (3) run the reference FiveStepPoserComputationProtocol
through GeneralPoser02 on the lambda_00 image and seeded poses, fp32 and fp64; (4) store a stride-3
pixel subset of all 33 outputs (+ the full posed frame of pose 0) in tests/golden/full_synth_io.npz.
"""
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import tha4.poser.modes.mode_07 as m07  # noqa: E402
from tha4.nn.common.unet import AttentionBlockArgs, UnetArgs  # noqa: E402
from tha4.nn.eyebrow_decomposer.eyebrow_decomposer_00 import EyebrowDecomposer00Args, EyebrowDecomposer00Factory  # noqa: E402
from tha4.nn.eyebrow_morphing_combiner.eyebrow_morphing_combiner_00 import (EyebrowMorphingCombiner00Args,  # noqa: E402
                                                                              EyebrowMorphingCombiner00Factory)
from tha4.nn.face_morpher.face_morpher_08 import FaceMorpher08Args, FaceMorpher08Factory  # noqa: E402
from tha4.nn.morpher.morpher_00 import Morpher00, Morpher00Args  # noqa: E402
from tha4.nn.nonlinearity_factory import ReLUFactory  # noqa: E402
from tha4.nn.normalization import InstanceNorm2dFactory  # noqa: E402
from tha4.nn.upscaler.upscaler_02 import Upscaler02, Upscaler02Args  # noqa: E402
from tha4.nn.util import BlockArgs  # noqa: E402
from tha4.poser.general_poser_02 import GeneralPoser02  # noqa: E402
from tha4.poser.modes.pose_parameters import get_pose_parameters  # noqa: E402

from oracle import full_oracle as fo  # noqa: E402
from oracle.student_oracle import random_poses  # noqa: E402

SUB = slice(1, None, 3)
SEED = 20260925
N_REF = 2


def build_reference_modules():
    def ba(inplace):
        return BlockArgs(initialization_method='he', use_spectral_norm=False,
                         normalization_layer_factory=InstanceNorm2dFactory(),
                         nonlinearity_factory=ReLUFactory(inplace=inplace))
    mods = {
        "eyebrow_decomposer": EyebrowDecomposer00Factory(EyebrowDecomposer00Args(128, 4, 64, 16, 6, 512, ba(True))).create(),
        "eyebrow_morphing_combiner": EyebrowMorphingCombiner00Factory(
            EyebrowMorphingCombiner00Args(128, 4, 12, 64, 16, 6, 512, ba(True))).create(),
        "face_morpher": FaceMorpher08Factory(FaceMorpher08Args(192, 4, 27, 64, 24, 6, 512, ba(False), True)).create(),
        "body_morpher": Morpher00(Morpher00Args(256, 4, 6, UnetArgs(
            4, 7, 64, [1, 2, 4, 4, 4], [False] * 4 + [True], 1, 4, None, 6, 256, AttentionBlockArgs(8, None, True), 0.0))),
        "upscaler": Upscaler02(Upscaler02Args(512, 4, 6, UnetArgs(
            4, 7, 32, [1, 2, 4, 8, 8, 8], [False] * 5 + [True], 1, 4, None, 6, 256, AttentionBlockArgs(8, None, True), 0.0))),
    }
    return mods


def main():
    w = fo.synth_full_weights(SEED)
    mods = build_reference_modules()
    for k, mod in mods.items():
        mod.load_state_dict({kk: torch.from_numpy(v) for kk, v in w[k].items()}, strict=True)
        mod.train(False)
    poser = GeneralPoser02(
        image_size=512, module_loaders={k: (lambda k=k: mods[k]) for k in mods},
        pose_parameters=get_pose_parameters().get_pose_parameter_groups(),
        output_list_func=m07.FiveStepPoserComputationProtocol(2).compute_func(),
        subrect=None, device=torch.device("cpu"), output_length=33, default_output_index=0)
    image = torch.from_numpy(np.load(os.path.join(HERE, "student_lambda_00_io.npz"))["image_f32"])
    poses = random_poses(4, seed=4321)
    io = {"poses": poses, "seed": np.int64(SEED)}
    torch.set_num_threads(8)
    ref32 = []
    with torch.no_grad():
        for i in range(N_REF):
            ref32.append([o[0].numpy().copy() for o in poser.get_posing_outputs(image, torch.from_numpy(poses[i]))])
    io["ref32_full_out0"] = ref32[0][0][None]
    for k in range(33):
        io[f"ref32_sub_out{k}"] = np.stack([r[k][:, SUB, SUB] for r in ref32])
    # fp64: modules .double() and default dtype fp64 (morpher_00.py:51 creates t with the default dtype)
    torch.set_default_dtype(torch.float64)
    for mod in mods.values():
        mod.double()
    poser64 = GeneralPoser02(
        image_size=512, module_loaders={k: (lambda k=k: mods[k]) for k in mods},
        pose_parameters=get_pose_parameters().get_pose_parameter_groups(),
        output_list_func=m07.FiveStepPoserComputationProtocol(2).compute_func(),
        subrect=None, device=torch.device("cpu"), output_length=33, default_output_index=0)
    with torch.no_grad():
        ref64 = [[o[0].numpy().copy() for o in poser64.get_posing_outputs(image.double(), torch.from_numpy(poses[i]).double())]
                 for i in range(1)]
    torch.set_default_dtype(torch.float32)
    for k in range(33):
        io[f"ref64_sub_out{k}"] = np.stack([r[k][:, SUB, SUB] for r in ref64]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "full_synth_io.npz"), **io)
    noise = {fo.OUTPUT_NAMES[k]: float(np.abs(ref32[0][k] - ref64[0][k]).max()) for k in range(33)}
    stats = {fo.OUTPUT_NAMES[k]: [float(ref32[0][k].min()), float(ref32[0][k].max())] for k in range(33)}
    with open(os.path.join(HERE, "full_synth_noise.json"), "w") as f:
        json.dump({"fp32_vs_fp64_maxabs": noise, "range": stats, "seed": SEED}, f, indent=1)
    print(json.dumps(noise, indent=1))
    print({k: v for k, v in stats.items() if "grid" in k})


if __name__ == "__main__":
    main()
