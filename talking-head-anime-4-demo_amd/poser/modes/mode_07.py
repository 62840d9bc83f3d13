"""Drop-in for the reference's ``tha4.poser.modes.mode_07`` (src/tha4/poser/modes/mode_07.py): the full
five-network THA4 system.  ``create_poser(device, module_file_names=None, eyebrow_morphed_image_index=2,
default_output_index=0)`` keeps the reference signature and defaults (:272-315): module keys are the
``Network`` enum names (:24-29), default files ``data/tha4/<name>.pt`` relative to the CWD (:279-293).
"""
from __future__ import annotations

from enum import Enum
from typing import Dict, Optional

import numpy as np
import torch

from ... import weights as _weights
from ..full_poser import HipFullPoser
from .pose_parameters import get_pose_parameters


class Network(Enum):                    # mode_07.py:24-33
    eyebrow_decomposer = 1
    eyebrow_morphing_combiner = 2
    face_morpher = 3
    body_morpher = 4
    upscaler = 5

    @property
    def outputs_key(self):
        return f"{self.name}_outputs"


NUM_EYEBROW_PARAMS = 12
NUM_FACE_PARAMS = 27
NUM_ROTATION_PARAMS = 6
EYEBROW_IMAGE_NO_COMBINE_ALPHA_INDEX = 2      # EyebrowMorphingCombiner00.EYEBROW_IMAGE_NO_COMBINE_ALPHA_INDEX


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
    return HipFullPoser(loaders, device, get_pose_parameters().get_pose_parameter_groups(), eyebrow_morphed_image_index,
                        default_output_index, max_batch, exact_fp32=exact_fp32, exact_decomposer=exact_decomposer)


def create_poser_from_state_dicts(device: torch.device, state_dicts: Dict[str, Dict[str, np.ndarray]],
                                  eyebrow_morphed_image_index: int = EYEBROW_IMAGE_NO_COMBINE_ALPHA_INDEX,
                                  default_output_index: int = 0, max_batch: int = 1, exact_fp32: bool = False, exact_decomposer: Optional[bool] = None) -> HipFullPoser:
    """Same poser from in-memory state_dicts keyed by the Network names (numpy or torch values)."""
    conv = {n: {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in sd.items()}
            for n, sd in state_dicts.items()}
    loaders = {net.name: (lambda n=net.name: conv[n]) for net in Network}
    return HipFullPoser(loaders, device, get_pose_parameters().get_pose_parameter_groups(), eyebrow_morphed_image_index,
                        default_output_index, max_batch, exact_fp32=exact_fp32, exact_decomposer=exact_decomposer)
