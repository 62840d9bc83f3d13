"""Drop-in for the reference's ``tha4.poser.modes.mode_12`` (src/tha4/poser/modes/mode_12.py): the first three
networks of the full system (eyebrow_decomposer -> eyebrow_morphing_combiner -> face_morpher), the teacher of the
face-morpher distillation (src/tha4/nn/siren/face_morpher/siren_face_morpher_00_trainer.py:23-26,
siren_face_morpher_protocols_00.py:83-89).  ``create_poser`` keeps the reference signature and defaults (:169-202).

Output list (mode_12.py:92-97): face_morpher 8 @192x192, eyebrow_morphing_combiner 8 @128x128, eyebrow_decomposer 6
@128x128 = 22 tensors; like the reference, ``get_output_length()`` reports the declared ``5 + 5 + 8 = 18`` (:200).
The same static schedule as mode_07 runs underneath (csrc/full_net.h), stopped after the face morpher.
"""
from __future__ import annotations

from enum import Enum
from typing import Dict, Optional

import numpy as np
import torch

from ... import weights as _weights
from ..full_poser import HipFullPoser
from .pose_parameters import get_pose_parameters


class Network(Enum):                    # mode_12.py:20-27
    eyebrow_decomposer = 1
    eyebrow_morphing_combiner = 2
    face_morpher = 3

    @property
    def outputs_key(self):
        return f"{self.name}_outputs"


NUM_EYEBROW_PARAMS = 12
NUM_FACE_PARAMS = 27
NUM_ROTATION_PARAMS = 6
EYEBROW_IMAGE_NO_COMBINE_ALPHA_INDEX = 2
DECLARED_OUTPUT_LENGTH = 5 + 5 + 8      # mode_12.py:200 (the list itself has 8 + 8 + 6 = 22 entries)
LIST_LENGTH = 8 + 8 + 6


def _make(loaders, device, eyebrow_morphed_image_index, default_output_index, max_batch, exact_fp32=False, exact_decomposer=None) -> HipFullPoser:
    p = HipFullPoser(loaders, device, get_pose_parameters().get_pose_parameter_groups(), eyebrow_morphed_image_index,
                     default_output_index, max_batch, exact_fp32=exact_fp32, exact_decomposer=exact_decomposer)
    p.num_networks = 3
    p.first_output = 11                 # face_morpher outputs are entries 11..18 of the mode_07 list (include/tha4_hip.h)
    p.list_length = LIST_LENGTH
    p.output_length = DECLARED_OUTPUT_LENGTH
    return p


def create_poser(device: torch.device,
                 module_file_names: Optional[Dict[str, str]] = None,
                 eyebrow_morphed_image_index: int = EYEBROW_IMAGE_NO_COMBINE_ALPHA_INDEX,
                 default_output_index: int = 0,
                 max_batch: int = 1, exact_fp32: bool = False, exact_decomposer: Optional[bool] = None) -> HipFullPoser:
    if module_file_names is None:
        module_file_names = {}
    for net in Network:
        if net.name not in module_file_names:
            module_file_names[net.name] = f"data/tha4/{net.name}.pt"
    loaders = {net.name: (lambda n=net.name: _weights.load_state_dict_file(module_file_names[n])) for net in Network}
    return _make(loaders, device, eyebrow_morphed_image_index, default_output_index, max_batch, exact_fp32, exact_decomposer)


def create_poser_from_state_dicts(device: torch.device, state_dicts: Dict[str, Dict[str, np.ndarray]],
                                  eyebrow_morphed_image_index: int = EYEBROW_IMAGE_NO_COMBINE_ALPHA_INDEX,
                                  default_output_index: int = 0, max_batch: int = 1, exact_fp32: bool = False, exact_decomposer: Optional[bool] = None) -> HipFullPoser:
    conv = {n.name: {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in state_dicts[n.name].items()}
            for n in Network}
    loaders = {net.name: (lambda n=net.name: conv[n]) for net in Network}
    return _make(loaders, device, eyebrow_morphed_image_index, default_output_index, max_batch, exact_fp32, exact_decomposer)
