"""The UNMODIFIED reference (imported read-only from /root/reference/src) as a CPU frame function.

*** TEST INFRASTRUCTURE ONLY *** - same rules as the other modules under oracle/: only tests/, the fixture scripts under
tests/golden/ and bench.py's `cpu_baseline` leg may import this.  /root/reference exists in the build container and NOT on the
GPU box: `available()` is False there and every caller falls back to the oracle's restatement ("port"), which equals this
bit for bit on every committed fixture (tests/test_oracle_golden.py, tests/test_full_oracle_golden.py).

  student_runner(name)   -> f(pose[45] float32 ndarray) -> list of 6 tensors    reference poser/modes/mode_14.py:134-162 +
                                                                                general_poser_02.py:57-79, shipped .pt / .png
  full_runner(weights)   -> f(image[4,512,512], pose[45]) -> list of 33 tensors reference modules (mode_07.py:137-269 factories)
                                                                                with the synthetic state_dicts, strict load
"""
from __future__ import annotations

import os
import sys

REF = "/root/reference"
REF_SRC = os.path.join(REF, "src")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_SRC, "tha4", "poser"))


def _import_reference():
    if not available():
        raise RuntimeError("the reference checkout (/root/reference) is not present on this machine")
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)


def student_runner(name: str = "lambda_00"):
    """The reference's own student poser on CPU for one shipped character; returns (run, image_f32)."""
    _import_reference()
    import PIL.Image
    import torch
    from tha4.poser.modes.mode_14 import create_poser  # reference, unmodified
    from tha4.shion.base.image_util import extract_pytorch_image_from_PIL_image
    cm = os.path.join(REF, "data/character_models", name)
    poser = create_poser(torch.device("cpu"), module_file_names={"face_morpher": os.path.join(cm, "face_morpher.pt"),
                                                                 "body_morpher": os.path.join(cm, "body_morpher.pt")})
    image = extract_pytorch_image_from_PIL_image(PIL.Image.open(os.path.join(cm, "character.png")))

    def run(pose):
        with torch.no_grad():
            return poser.get_posing_outputs(image, torch.as_tensor(pose, dtype=torch.float32))
    return run, image


def build_reference_full_modules():
    """The five networks of mode_07 from the reference's own factories and arguments (mode_07.py:137-269), without torch_load."""
    _import_reference()
    from tha4.nn.common.unet import AttentionBlockArgs, UnetArgs
    from tha4.nn.eyebrow_decomposer.eyebrow_decomposer_00 import EyebrowDecomposer00Args, EyebrowDecomposer00Factory
    from tha4.nn.eyebrow_morphing_combiner.eyebrow_morphing_combiner_00 import EyebrowMorphingCombiner00Args, EyebrowMorphingCombiner00Factory
    from tha4.nn.face_morpher.face_morpher_08 import FaceMorpher08Args, FaceMorpher08Factory
    from tha4.nn.morpher.morpher_00 import Morpher00, Morpher00Args
    from tha4.nn.nonlinearity_factory import ReLUFactory
    from tha4.nn.normalization import InstanceNorm2dFactory
    from tha4.nn.upscaler.upscaler_02 import Upscaler02, Upscaler02Args
    from tha4.nn.util import BlockArgs

    def ba(inplace):
        return BlockArgs(initialization_method='he', use_spectral_norm=False,
                         normalization_layer_factory=InstanceNorm2dFactory(),
                         nonlinearity_factory=ReLUFactory(inplace=inplace))
    return {
        "eyebrow_decomposer": EyebrowDecomposer00Factory(EyebrowDecomposer00Args(128, 4, 64, 16, 6, 512, ba(True))).create(),
        "eyebrow_morphing_combiner": EyebrowMorphingCombiner00Factory(
            EyebrowMorphingCombiner00Args(128, 4, 12, 64, 16, 6, 512, ba(True))).create(),
        "face_morpher": FaceMorpher08Factory(FaceMorpher08Args(192, 4, 27, 64, 24, 6, 512, ba(False), True)).create(),
        "body_morpher": Morpher00(Morpher00Args(256, 4, 6, UnetArgs(
            4, 7, 64, [1, 2, 4, 4, 4], [False] * 4 + [True], 1, 4, None, 6, 256, AttentionBlockArgs(8, None, True), 0.0))),
        "upscaler": Upscaler02(Upscaler02Args(512, 4, 6, UnetArgs(
            4, 7, 32, [1, 2, 4, 8, 8, 8], [False] * 5 + [True], 1, 4, None, 6, 256, AttentionBlockArgs(8, None, True), 0.0))),
    }


def full_runner(weights):
    """Reference GeneralPoser02 + FiveStepPoserComputationProtocol over the reference modules loaded (strict) with `weights`
    ({network: {key: ndarray}}); returns run(image, pose) -> the reference's list of 33 outputs."""
    _import_reference()
    import torch
    import tha4.poser.modes.mode_07 as m07
    from tha4.poser.general_poser_02 import GeneralPoser02
    from tha4.poser.modes.pose_parameters import get_pose_parameters
    mods = build_reference_full_modules()
    for k, mod in mods.items():
        mod.load_state_dict({kk: torch.as_tensor(v) for kk, v in weights[k].items()}, strict=True)
        mod.train(False)
    poser = GeneralPoser02(
        image_size=512, module_loaders={k: (lambda k=k: mods[k]) for k in mods},
        pose_parameters=get_pose_parameters().get_pose_parameter_groups(),
        output_list_func=m07.FiveStepPoserComputationProtocol(2).compute_func(),
        subrect=None, device=torch.device("cpu"), output_length=33, default_output_index=0)

    def run(image, pose):
        with torch.no_grad():
            return poser.get_posing_outputs(torch.as_tensor(image, dtype=torch.float32), torch.as_tensor(pose, dtype=torch.float32))
    return run
