"""The 45-slot pose vector layout (reference: src/tha4/poser/modes/pose_parameters.py:4-36).

slots 0-11 eyebrow (6 groups x L/R), 12-23 eye (6 x L/R), 24-25 iris_small, 26-31 mouth
aaa/iii/uuu/eee/ooo/delta, 32-35 mouth corners (2 x L/R), 36 smirk, 37-38 iris rotation x/y,
39-41 head_x/head_y/neck_z, 42-43 body_y/body_z, 44 breathing.
"""
from ..poser import PoseParameterCategory as Cat
from ..poser import PoseParameters

_PAIRS_EYEBROW = ["eyebrow_troubled", "eyebrow_angry", "eyebrow_lowered", "eyebrow_raised", "eyebrow_happy",
                  "eyebrow_serious"]
_PAIRS_EYE = ["eye_wink", "eye_happy_wink", "eye_surprised", "eye_relaxed", "eye_unimpressed",
              "eye_raised_lower_eyelid"]
_MOUTH_SINGLE = ["mouth_aaa", "mouth_iii", "mouth_uuu", "mouth_eee", "mouth_ooo", "mouth_delta"]
_SIGNED = [("iris_rotation_x", Cat.IRIS_ROTATION), ("iris_rotation_y", Cat.IRIS_ROTATION),
           ("head_x", Cat.FACE_ROTATION), ("head_y", Cat.FACE_ROTATION), ("neck_z", Cat.FACE_ROTATION),
           ("body_y", Cat.BODY_ROTATION), ("body_z", Cat.BODY_ROTATION)]


def get_pose_parameters() -> PoseParameters:
    b = PoseParameters.Builder()
    for name in _PAIRS_EYEBROW:
        b.add_parameter_group(name, Cat.EYEBROW, arity=2)
    for name in _PAIRS_EYE:
        b.add_parameter_group(name, Cat.EYE, arity=2)
    b.add_parameter_group("iris_small", Cat.IRIS_MORPH, arity=2)
    for name in _MOUTH_SINGLE:
        b.add_parameter_group(name, Cat.MOUTH, arity=1, default_value=1.0 if name == "mouth_aaa" else 0.0)
    b.add_parameter_group("mouth_lowered_corner", Cat.MOUTH, arity=2)
    b.add_parameter_group("mouth_raised_corner", Cat.MOUTH, arity=2)
    b.add_parameter_group("mouth_smirk", Cat.MOUTH, arity=1)
    for name, cat in _SIGNED:
        b.add_parameter_group(name, cat, arity=1, range=(-1.0, 1.0))
    b.add_parameter_group("breathing", Cat.BREATHING, arity=1, range=(0.0, 1.0))
    return b.build()
