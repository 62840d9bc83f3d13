"""``Poser`` implementation for the distilled student (mode_14) on top of the HIP C ABI.

Mirrors the surface of the reference's ``GeneralPoser02`` (src/tha4/poser/general_poser_02.py:10-98)
as used by ``mode_14.create_poser`` (src/tha4/poser/modes/mode_14.py:134-162):

  * modules are loaded lazily on the first call (general_poser_02.py:41-49);
  * a 3-D image / 1-D pose is promoted to a batch of one (:66-69); results are always batched;
  * ``pose(image, pose, output_index=None)`` returns ``get_posing_outputs(...)[output_index]``
    with ``None`` -> ``default_output_index`` (:57-61);
  * ``get_posing_outputs`` returns the 6 tensors of TwoStepPoserComputationProtocol's
    "all_outputs" in the reference order (mode_14.py:85-88):
        [blended, alpha, color_change, warped, grid_change, face_morpher_output];
  * ``to(device)``, ``free()``, ``get_dtype()``, ``get_image_size()``, ``get_output_length()``,
    ``get_pose_parameter_groups()``, ``get_num_parameters()`` as in the reference.

All work is enqueued on ``torch.cuda.current_stream(device)`` without synchronising, so callers can
bracket ``pose()`` with ``torch.cuda.Event`` exactly like full_manual_poser.py:388-398.
PyTorch is used for device memory and streams only; the arithmetic happens in libtha4_hip.so.
There is no CPU path: a non-CUDA device raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from .. import _capi
from .poser import PoseParameterGroup, Poser

IMAGE_SIZE = 512
FACE_SIZE = 128
NUM_OUTPUTS = 6


def aten_position_axes() -> Dict[int, np.ndarray]:
    """fp32 affine_grid axes as the local PyTorch build produces them on CPU - what the reference
    feeds its SIREN stacks every frame (siren_morpher_03.py:92-99).  Constants, computed once."""
    import torch.nn.functional as F
    out = {}
    ident = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]).unsqueeze(0)
    for s in (128, 256, 512):
        out[s] = F.affine_grid(ident, [1, 1, s, s], align_corners=False)[0, 0, :, 0].numpy().copy()
    return out


class HipStudentPoser(Poser):
    def __init__(self,
                 state_dict_loaders: Dict[str, Callable[[], Dict[str, np.ndarray]]],
                 device: torch.device,
                 pose_parameters: List[PoseParameterGroup],
                 default_output_index: int = 0,
                 max_batch: int = 1,
                 position_axes: Optional[Dict[int, np.ndarray]] = None,
                 exact_fp32: bool = False,
                 dtype: torch.dtype = torch.float):
        self.state_dict_loaders = state_dict_loaders
        self.device = torch.device(device)
        self.pose_parameters = pose_parameters
        self.default_output_index = default_output_index
        self.image_size = IMAGE_SIZE
        self.output_length = NUM_OUTPUTS
        self.dtype = dtype
        self.num_parameters = sum(p.get_arity() for p in pose_parameters)
        self.position_axes = position_axes
        self.exact_fp32 = bool(exact_fp32)   # A/B switch: fp32-product MFMA kernels instead of the fp16 hi/lo split
        self._max_batch = max(1, int(max_batch))
        self._state_dicts = None
        self._lib = None
        self._handle = None

    # ---- reference surface -----------------------------------------------------------------
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
        """Lazy load (general_poser_02.py:41-49): read the state_dicts and create the native handle."""
        self._ensure_handle(self._max_batch)
        return self._state_dicts

    def pose(self, image: Tensor, pose: Tensor, output_index: Optional[int] = None, out: Optional[Tensor] = None) -> Tensor:
        """Reference signature plus an optional ``out`` (extension; output 0 only): a contiguous float32 ``[B,4,512,512]``
        device tensor the posed frames are written into directly - e.g. a row block of a gather buffer - instead of a
        fresh allocation."""
        if output_index is None:
            output_index = self.default_output_index
        if output_index == 0:
            return self._run(image, pose, all_outputs=False, out=out)[0]
        if out is not None:
            raise AssertionError("`out` is only supported for output_index 0")
        return self.get_posing_outputs(image, pose)[output_index]

    def get_posing_outputs(self, image: Tensor, pose: Tensor) -> List[Tensor]:
        return self._run(image, pose, all_outputs=True)

    def free(self):
        self._destroy_handle()
        self._state_dicts = None

    def set_state_dicts(self, face_state_dict: Dict[str, np.ndarray], body_state_dict: Dict[str, np.ndarray]):
        """Hot-swap the character: new student weights into the LIVE native handle -
        ``tha4_student_set_weights`` overwrites the packed parameter blob in place, nothing is re-allocated (workspace,
        max_batch, position axes stay).  Before the first call it just replaces what the lazy loaders will deliver."""
        conv = lambda sd: {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in sd.items()}
        face, body = conv(face_state_dict), conv(body_state_dict)
        if self._handle is not None:
            # the native call validates and packs first: a rejected pair ("not a mode_14 student") leaves the live handle AND
            # the Python-side state (loaders, get_modules(), a later re-create) on the old character
            weights, keep = _capi.build_student_weights(face, body)
            st = self._lib.tha4_student_set_weights(self._handle, C.byref(weights))
            _capi.check(self._lib, st, "tha4_student_set_weights")
            del keep
        self.state_dict_loaders = {"face_morpher": lambda: face, "body_morpher": lambda: body}
        self._state_dicts = {"face_morpher": face, "body_morpher": body}
        return self

    def load_character(self, module_file_names: Dict[str, str]):
        """``set_state_dicts`` from two reference ``.pt`` files (keys "face_morpher" / "body_morpher", mode_14.py:14-15)."""
        from .. import weights as _weights
        return self.set_state_dicts(_weights.load_state_dict_file(module_file_names["face_morpher"]),
                                    _weights.load_state_dict_file(module_file_names["body_morpher"]))

    def to(self, device: torch.device) -> "HipStudentPoser":
        device = torch.device(device)
        if device == self.device:
            return self
        self._destroy_handle()
        self.device = device
        return self

    # ---- native plumbing ---------------------------------------------------------------------
    def _device_index(self) -> int:
        if self.device.type != "cuda":
            raise _capi.Tha4Error(f"HipStudentPoser needs a ROCm GPU device, got {self.device} (no CPU path exists)")
        return self.device.index if self.device.index is not None else torch.cuda.current_device()

    def _destroy_handle(self):
        if self._handle is not None and self._lib is not None:
            self._lib.tha4_student_destroy(self._handle)
        self._handle = None

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
        weights, keep = _capi.build_student_weights(self._state_dicts["face_morpher"], self._state_dicts["body_morpher"])
        axes, keep2 = _capi.build_position_axes(self.position_axes)
        handle = C.c_void_p()
        st = self._lib.tha4_student_create_ex(C.byref(weights), C.byref(axes) if axes is not None else None, dev,
                                              self._max_batch, _capi.STUDENT_EXACT_FP32 if self.exact_fp32 else 0,
                                              C.byref(handle))
        _capi.check(self._lib, st, "tha4_student_create_ex")
        del keep, keep2
        self._handle = handle

    def _check_inputs(self, image: Tensor, pose: Tensor):
        if image.dim() == 3:
            image = image.unsqueeze(0)
        if pose.dim() == 1:
            pose = pose.unsqueeze(0)
        if image.dim() != 4 or tuple(image.shape[1:]) != (4, IMAGE_SIZE, IMAGE_SIZE):
            raise AssertionError(f"image must be [B,4,512,512] or [4,512,512], got {tuple(image.shape)}")
        if pose.dim() != 2 or pose.shape[1] != self.num_parameters:
            raise AssertionError(f"pose must be [B,{self.num_parameters}] or [{self.num_parameters}], got {tuple(pose.shape)}")
        b = pose.shape[0]
        if image.shape[0] not in (1, b):
            raise AssertionError(f"image batch {image.shape[0]} does not match pose batch {b}")
        for name, t in (("image", image), ("pose", pose)):
            if t.dtype != torch.float32:
                raise AssertionError(f"{name} must be float32, got {t.dtype}")
            if t.device != self.device and not (t.device.type == "cuda" and self.device.type == "cuda"
                                                and t.device.index == self._device_index()):
                raise AssertionError(f"{name} is on {t.device}, poser is on {self.device}")
        return image.contiguous(), pose.contiguous(), b

    def _run(self, image: Tensor, pose: Tensor, all_outputs: bool, out: Optional[Tensor] = None) -> List[Tensor]:
        image, pose, b = self._check_inputs(image, pose)
        self._ensure_handle(b)
        dev = self._device_index()
        opts = dict(dtype=torch.float32, device=image.device)
        if out is not None:
            if (tuple(out.shape) != (b, 4, IMAGE_SIZE, IMAGE_SIZE) or out.dtype != torch.float32 or out.device != image.device
                    or not out.is_contiguous()):
                raise AssertionError(f"out must be a contiguous float32 [{b},4,{IMAGE_SIZE},{IMAGE_SIZE}] tensor on {image.device}")
            blended = out
        else:
            blended = torch.empty((b, 4, IMAGE_SIZE, IMAGE_SIZE), **opts)
        outs = [blended]
        aux_ref = None
        if all_outputs:
            alpha = torch.empty((b, 1, IMAGE_SIZE, IMAGE_SIZE), **opts)
            color = torch.empty((b, 4, IMAGE_SIZE, IMAGE_SIZE), **opts)
            warped = torch.empty((b, 4, IMAGE_SIZE, IMAGE_SIZE), **opts)
            grid = torch.empty((b, 2, IMAGE_SIZE, IMAGE_SIZE), **opts)
            face = torch.empty((b, 4, FACE_SIZE, FACE_SIZE), **opts)
            aux = _capi.Tha4StudentAux(alpha.data_ptr(), color.data_ptr(), warped.data_ptr(), grid.data_ptr(),
                                       face.data_ptr())
            aux_ref = C.byref(aux)
            outs += [alpha, color, warped, grid, face]
        stride = 0 if (image.shape[0] == 1 and b > 1) else 4 * IMAGE_SIZE * IMAGE_SIZE
        stream = torch.cuda.current_stream(dev).cuda_stream
        st = self._lib.tha4_student_pose(self._handle, image.data_ptr(), stride, pose.data_ptr(), b,
                                         blended.data_ptr(), aux_ref, C.c_void_p(stream))
        _capi.check(self._lib, st, "tha4_student_pose")
        return outs

    def pose_display_rgba8(self, image: Tensor, pose: Tensor, background_rgb=None, out: Optional[Tensor] = None,
                           want_frame: bool = False):
        """``pose()`` followed by the display post-processing every puppeteer runs on the device
        (character_model_ifacialmocap_puppeteer.py:325-349,377-381: clip((x+1)/2) -> linear->sRGB -> optional background
        blend -> HWC -> *255 -> ``.byte()``), FUSED into the kernel that composes the frame (tha4_display): returns the
        ``uint8 [B,512,512,4]`` frame; the fp32 frame is neither written nor re-read unless ``want_frame`` asks for it
        too (then ``(rgba8, frame)`` is returned)."""
        image, pose, b = self._check_inputs(image, pose)
        self._ensure_handle(b)
        dev = self._device_index()
        if out is not None:
            if (tuple(out.shape) != (b, IMAGE_SIZE, IMAGE_SIZE, 4) or out.dtype != torch.uint8 or out.device != image.device
                    or not out.is_contiguous()):
                raise AssertionError(f"out must be a contiguous uint8 [{b},{IMAGE_SIZE},{IMAGE_SIZE},4] tensor on {image.device}")
            rgba = out
        else:
            rgba = torch.empty((b, IMAGE_SIZE, IMAGE_SIZE, 4), dtype=torch.uint8, device=image.device)
        frame = torch.empty((b, 4, IMAGE_SIZE, IMAGE_SIZE), dtype=torch.float32, device=image.device) if want_frame else None
        aux = _capi.Tha4StudentAux()
        aux.display, keep = _capi.make_display(rgba.data_ptr(), background_rgb)
        stride = 0 if (image.shape[0] == 1 and b > 1) else 4 * IMAGE_SIZE * IMAGE_SIZE
        stream = torch.cuda.current_stream(dev).cuda_stream
        st = self._lib.tha4_student_pose(self._handle, image.data_ptr(), stride, pose.data_ptr(), b,
                                         frame.data_ptr() if want_frame else None, C.byref(aux), C.c_void_p(stream))
        _capi.check(self._lib, st, "tha4_student_pose")
        del keep
        return (rgba, frame) if want_frame else rgba

    def debug_hand_off(self, level: int, frame: int = 0) -> Tensor:
        """Test hook (tha4_student_debug_read): the z hand-off image the most recent pose call wrote for body level
        ``level`` (1 or 2) and batch slot ``frame``, un-permuted to ``[C_padded, S, S]`` fp32 on the CPU (S = 128 / 256) -
        divided by the factor the kernels fold into it (``tha4_student_hand_off_scale``: omega_0 / 2 pi), i.e. equal to
        ``oracle.student_intermediates(...)["z<level>"]`` on the real channels."""
        nb, side = (12, 128) if level == 1 else (6, 256)
        host = torch.empty(nb * 16 * side * side, dtype=torch.float32)
        _capi.check(self._lib, self._lib.tha4_student_debug_read(self._handle, level - 1, frame, host.data_ptr()),
                    "tha4_student_debug_read")
        scale = float(self._lib.tha4_student_hand_off_scale(self._handle))
        return host.view(nb, 4, side * side, 4).permute(0, 1, 3, 2).reshape(nb * 16, side, side) / scale   # [block][g][pixel][j] -> channel 16b+4g+j

    # ---- measurement hooks (bench.py) ----------------------------------------------------------
    def set_timing(self, enable: bool):
        self._ensure_handle(self._max_batch)
        _capi.check(self._lib, self._lib.tha4_student_set_timing(self._handle, int(enable)), "tha4_student_set_timing")

    def last_kernel_ms(self, kernel: int) -> float:
        ms = C.c_float()
        _capi.check(self._lib, self._lib.tha4_student_last_ms(self._handle, kernel, C.byref(ms)), "tha4_student_last_ms")
        return float(ms.value)
