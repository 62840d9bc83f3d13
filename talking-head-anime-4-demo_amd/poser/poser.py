"""Mirror of the reference plugin boundary ``tha4.poser.poser`` (src/tha4/poser/poser.py).

Same class names, constructor arguments, method names and return conventions, written from the
interface description so that apps written against the reference
(full_manual_poser, character_model_* puppeteers) keep working when handed one of these objects.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from enum import Enum
from typing import List, Optional, Tuple

import torch
from torch import Tensor


class PoseParameterCategory(Enum):          # poser.py:8-16
    EYEBROW = 1
    EYE = 2
    IRIS_MORPH = 3
    IRIS_ROTATION = 4
    MOUTH = 5
    FACE_ROTATION = 6
    BODY_ROTATION = 7
    BREATHING = 8


class PoseParameterGroup:                   # poser.py:20-69
    """One named slider (arity 1) or left/right pair (arity 2) of the 45-float pose vector."""

    def __init__(self, group_name: str, parameter_index: int, category: PoseParameterCategory, arity: int = 1,
                 discrete: bool = False, default_value: float = 0.0, range: Optional[Tuple[float, float]] = None):
        if arity not in (1, 2):
            raise AssertionError("arity must be 1 or 2")
        self.group_name = group_name
        self.parameter_index = parameter_index
        self.category = category
        self.arity = arity
        self.discrete = discrete
        self.default_value = default_value
        self.range = (0.0, 1.0) if range is None else range
        self.parameter_names = [group_name] if arity == 1 else [group_name + "_left", group_name + "_right"]

    def get_arity(self) -> int:
        return self.arity

    def get_group_name(self) -> str:
        return self.group_name

    def get_parameter_names(self) -> List[str]:
        return self.parameter_names

    def is_discrete(self) -> bool:
        return self.discrete

    def get_range(self) -> Tuple[float, float]:
        return self.range

    def get_default_value(self):
        return self.default_value

    def get_parameter_index(self):
        return self.parameter_index

    def get_category(self) -> PoseParameterCategory:
        return self.category


class PoseParameters:                       # poser.py:72-129
    def __init__(self, pose_parameter_groups: List[PoseParameterGroup]):
        self.pose_parameter_groups = pose_parameter_groups

    def _names(self) -> List[str]:
        return [n for g in self.pose_parameter_groups for n in g.get_parameter_names()]

    def get_parameter_index(self, name: str) -> int:
        names = self._names()
        if name not in names:
            raise RuntimeError("Cannot find parameter with name %s" % name)
        return names.index(name)

    def get_parameter_name(self, index: int) -> str:
        names = self._names()
        assert 0 <= index < len(names)
        return names[index]

    def get_pose_parameter_groups(self):
        return self.pose_parameter_groups

    def get_parameter_count(self):
        return len(self._names())

    class Builder:
        def __init__(self):
            self.index = 0
            self.pose_parameter_groups: List[PoseParameterGroup] = []

        def add_parameter_group(self, group_name: str, category: PoseParameterCategory, arity: int = 1,
                                discrete: bool = False, default_value: float = 0.0,
                                range: Optional[Tuple[float, float]] = None):
            self.pose_parameter_groups.append(
                PoseParameterGroup(group_name, self.index, category, arity, discrete, default_value, range))
            self.index += arity
            return self

        def build(self) -> "PoseParameters":
            return PoseParameters(self.pose_parameter_groups)


class Poser(ABC):                           # poser.py:132-162
    @abstractmethod
    def get_image_size(self) -> int:
        ...

    @abstractmethod
    def get_output_length(self) -> int:
        ...

    @abstractmethod
    def get_pose_parameter_groups(self) -> List[PoseParameterGroup]:
        ...

    @abstractmethod
    def get_num_parameters(self) -> int:
        ...

    @abstractmethod
    def pose(self, image: Tensor, pose: Tensor, output_index: int = 0) -> Tensor:
        ...

    @abstractmethod
    def get_posing_outputs(self, image: Tensor, pose: Tensor) -> List[Tensor]:
        ...

    def get_dtype(self) -> torch.dtype:
        return torch.float

    @abstractmethod
    def to(self, device: torch.device):
        ...
