"""Drop-in for the reference's ``tha4.poser.modes.mode_14`` (src/tha4/poser/modes/mode_14.py).

``create_poser(device, module_file_names=None, default_output_index=0)`` has the reference's
signature and defaults (mode_14.py:134-162): the two module keys are ``"face_morpher"`` and
``"body_morpher"`` (:14-15) and the default files are the shipped lambda_00 student, relative to the
current working directory (:140-145).  The returned object implements the reference's ``Poser``
interface; the computation runs in the hand-written gfx950 kernels behind include/tha4_hip.h.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from ... import weights as _weights
from ..student_poser import HipStudentPoser, aten_position_axes
from .pose_parameters import get_pose_parameters

KEY_FACE_MORPHER = "face_morpher"
KEY_BODY_MORPHER = "body_morpher"

DEFAULT_FACE_MORPHER_FILE = "data/character_models/lambda_00/face_morpher.pt"
DEFAULT_BODY_MORPHER_FILE = "data/character_models/lambda_00/body_morpher.pt"


def load_face_morpher(file_name: str) -> Dict[str, np.ndarray]:
    """State-dict loader for the face SIREN (reference builds the nn.Module here, mode_14.py:93-106)."""
    return _weights.load_state_dict_file(file_name)


def load_body_morpher(file_name: str) -> Dict[str, np.ndarray]:
    """State-dict loader for the 3-level body SIREN (mode_14.py:109-131)."""
    return _weights.load_state_dict_file(file_name)


def create_poser(device: torch.device,
                 module_file_names: Optional[Dict[str, str]] = None,
                 default_output_index: int = 0,
                 max_batch: int = 1,
                 match_aten_positions: bool = True,
                 exact_fp32: bool = False) -> HipStudentPoser:
    """Same contract as the reference factory; ``max_batch`` (workspace pre-sizing, grows on demand)
    and ``match_aten_positions`` (use the local ATen affine_grid fp32 axes instead of the exact dyadic
    ones, see include/tha4_hip.h) are additions with reference-compatible defaults."""
    if module_file_names is None:
        module_file_names = {}
    if KEY_FACE_MORPHER not in module_file_names:
        module_file_names[KEY_FACE_MORPHER] = DEFAULT_FACE_MORPHER_FILE
    if KEY_BODY_MORPHER not in module_file_names:
        module_file_names[KEY_BODY_MORPHER] = DEFAULT_BODY_MORPHER_FILE
    loaders = {
        KEY_FACE_MORPHER: lambda: load_face_morpher(module_file_names[KEY_FACE_MORPHER]),
        KEY_BODY_MORPHER: lambda: load_body_morpher(module_file_names[KEY_BODY_MORPHER]),
    }
    return HipStudentPoser(
        state_dict_loaders=loaders,
        device=device,
        pose_parameters=get_pose_parameters().get_pose_parameter_groups(),
        default_output_index=default_output_index,
        max_batch=max_batch,
        position_axes=aten_position_axes() if match_aten_positions else None,
        exact_fp32=exact_fp32)


def create_poser_from_state_dicts(device: torch.device,
                                  face_state_dict: Dict[str, np.ndarray],
                                  body_state_dict: Dict[str, np.ndarray],
                                  default_output_index: int = 0,
                                  max_batch: int = 1,
                                  match_aten_positions: bool = True,
                                  exact_fp32: bool = False) -> HipStudentPoser:
    """Same poser from in-memory state_dicts (numpy or torch values, reference key layout)."""
    face = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in face_state_dict.items()}
    body = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in body_state_dict.items()}
    loaders = {KEY_FACE_MORPHER: lambda: face, KEY_BODY_MORPHER: lambda: body}
    return HipStudentPoser(
        state_dict_loaders=loaders,
        device=device,
        pose_parameters=get_pose_parameters().get_pose_parameter_groups(),
        default_output_index=default_output_index,
        max_batch=max_batch,
        position_axes=aten_position_axes() if match_aten_positions else None,
        exact_fp32=exact_fp32)
