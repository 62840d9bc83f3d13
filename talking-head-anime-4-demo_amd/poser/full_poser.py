"""``Poser`` implementation for the full THA4 system (mode_07) on top of the HIP C ABI.

Mirrors ``GeneralPoser02`` as configured by ``mode_07.create_poser`` (src/tha4/poser/modes/mode_07.py:272-315):
33 outputs in the reference order (:126-132), ``output_index`` selection, lazy loading, batching of
3-D image / 1-D pose, ``to`` / ``free``.  The eyebrow-decomposer cache of the reference
(``FiveStepPoserComputationProtocol.compute_func``, :54-70: reuse while ``max|image - cached| == 0``)
is kept, but keyed on the tensor's identity (data_ptr, _version, shape) instead of a device->host
``.item()`` synchronisation; pass ``image_changed=True`` to force a refresh after an in-place edit
that does not bump ``_version``.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from .. import _capi
from .poser import PoseParameterGroup, Poser

OUT_CHANNELS = [4, 1, 4, 2, 4, 4, 4, 1, 4, 2, 4, 4, 1, 4, 4, 1, 4, 4, 2, 4, 1, 4, 4, 1, 4, 4, 2, 4, 1, 4, 4, 1, 4]
OUT_SIZE = [512] * 6 + [256] * 5 + [192] * 8 + [128] * 14


class HipFullPoser(Poser):
    def __init__(self,
                 state_dict_loaders: Dict[str, Callable[[], Dict[str, np.ndarray]]],
                 device: torch.device,
                 pose_parameters: List[PoseParameterGroup],
                 eyebrow_morphed_image_index: int = 2,
                 default_output_index: int = 0,
                 max_batch: int = 1,
                 dtype: torch.dtype = torch.float):
        self.state_dict_loaders = state_dict_loaders
        self.device = torch.device(device)
        self.pose_parameters = pose_parameters
        self.eyebrow_morphed_image_index = eyebrow_morphed_image_index
        self.default_output_index = default_output_index
        self.image_size = 512
        self.output_length = _capi.FULL_NUM_OUTPUTS
        self.dtype = dtype
        self.num_parameters = sum(p.get_arity() for p in pose_parameters)
        self._max_batch = max(1, int(max_batch))
        self._state_dicts = None
        self._lib = None
        self._handle = None
        self._cache_key = None

    def get_image_size(self) -> int:
        return self.image_size

    def get_output_length(self) -> int:
        return self.output_length

    def get_pose_parameter_groups(self) -> List[PoseParameterGroup]:
        return self.pose_parameters

    def get_num_parameters(self) -> int:
        return self.num_parameters

    def get_dtype(self) -> torch.dtype:
        return self.dtype

    def get_modules(self):
        self._ensure_handle(self._max_batch)
        return self._state_dicts

    def pose(self, image: Tensor, pose: Tensor, output_index: Optional[int] = None, image_changed: bool = False) -> Tensor:
        if output_index is None:
            output_index = self.default_output_index
        return self._run(image, pose, [output_index], image_changed)[0]

    def get_posing_outputs(self, image: Tensor, pose: Tensor, image_changed: bool = False) -> List[Tensor]:
        return self._run(image, pose, list(range(self.output_length)), image_changed)

    def free(self):
        self._destroy_handle()
        self._state_dicts = None

    def to(self, device: torch.device) -> "HipFullPoser":
        device = torch.device(device)
        if device == self.device:
            return self
        self._destroy_handle()
        self.device = device
        return self

    # ---- native plumbing ---------------------------------------------------------------------
    def _device_index(self) -> int:
        if self.device.type != "cuda":
            raise _capi.Tha4Error(f"HipFullPoser needs a ROCm GPU device, got {self.device} (no CPU path exists)")
        return self.device.index if self.device.index is not None else torch.cuda.current_device()

    def _destroy_handle(self):
        if self._handle is not None and self._lib is not None:
            self._lib.tha4_full_destroy(self._handle)
        self._handle = None
        self._cache_key = None

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:
            pass

    def _ensure_handle(self, batch: int):
        if self._handle is not None and batch <= self._max_batch:
            return
        dev = self._device_index()
        if self._lib is None:
            self._lib = _capi.load_library()
        if self._state_dicts is None:
            self._state_dicts = {k: loader() for k, loader in self.state_dict_loaders.items()}
        self._destroy_handle()
        self._max_batch = max(self._max_batch, batch)
        weights, keep = _capi.build_full_weights(self._state_dicts)
        handle = C.c_void_p()
        st = self._lib.tha4_full_create(C.byref(weights), self.eyebrow_morphed_image_index, dev, self._max_batch, C.byref(handle))
        _capi.check(self._lib, st, "tha4_full_create")
        del keep
        self._handle = handle

    def _run(self, image: Tensor, pose: Tensor, wanted: List[int], image_changed: bool) -> List[Tensor]:
        if image.dim() == 3:
            image = image.unsqueeze(0)
        if pose.dim() == 1:
            pose = pose.unsqueeze(0)
        if image.dim() != 4 or tuple(image.shape[1:]) != (4, 512, 512):
            raise AssertionError(f"image must be [B,4,512,512] or [4,512,512], got {tuple(image.shape)}")
        if pose.dim() != 2 or pose.shape[1] != self.num_parameters:
            raise AssertionError(f"pose must be [B,{self.num_parameters}], got {tuple(pose.shape)}")
        b = pose.shape[0]
        if image.shape[0] not in (1, b):
            raise AssertionError(f"image batch {image.shape[0]} does not match pose batch {b}")
        if image.dtype != torch.float32 or pose.dtype != torch.float32:
            raise AssertionError("image and pose must be float32")
        image, pose = image.contiguous(), pose.contiguous()
        self._ensure_handle(b)
        dev = self._device_index()
        key = (image.data_ptr(), image._version, tuple(image.shape), b)
        reuse = (not image_changed) and key == self._cache_key
        outs = {}
        ptrs = (C.c_void_p * _capi.FULL_NUM_OUTPUTS)()
        for i in sorted(set(wanted) | {0}):
            t = torch.empty((b, OUT_CHANNELS[i], OUT_SIZE[i], OUT_SIZE[i]), dtype=torch.float32, device=image.device)
            outs[i] = t
            ptrs[i] = t.data_ptr()
        stride = 0 if (image.shape[0] == 1 and b > 1) else 4 * 512 * 512
        stream = torch.cuda.current_stream(dev).cuda_stream
        st = self._lib.tha4_full_pose(self._handle, image.data_ptr(), stride, pose.data_ptr(), b, ptrs, int(reuse),
                                      C.c_void_p(stream))
        _capi.check(self._lib, st, "tha4_full_pose")
        self._cache_key = key
        return [outs[i] for i in wanted]
