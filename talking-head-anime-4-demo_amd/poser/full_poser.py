"""``Poser`` implementation for the full THA4 system (mode_07) on top of the HIP C ABI.

Mirrors ``GeneralPoser02`` as configured by ``mode_07.create_poser`` (src/tha4/poser/modes/mode_07.py:272-315):
33 outputs in the reference order (:126-132), ``output_index`` selection, lazy loading, batching of
3-D image / 1-D pose, ``to`` / ``free``.  The eyebrow-decomposer cache of the reference
(``FiveStepPoserComputationProtocol.compute_func``, :54-70: reuse while ``max|image - cached| == 0``)
is kept without its device->host ``.item()`` synchronisation.  The reference decides reuse by CONTENT and keeps a
reference to the cached batch (``cached_batch_0``); here reuse is decided, in this order, by
  * ``image_version=<int>`` given by the caller (SURVEY.md §8b proposal): reuse iff the version, shape and batch equal
    the previous call's - the caller vouches for the content;
  * otherwise the tensor's storage identity ``(data_ptr, shape, strides, _version)`` while the poser holds a STRONG
    reference to the cached tensor, so the allocator cannot hand its address to a different image, and in-place
    edits bump ``_version``.  Nothing is cached when the input had to be copied (non-contiguous) or does not track a
    version (inference tensors): those calls are always "cold", which is the safe side;
  * ``image_changed=True`` forces a refresh.
``content_cache = True`` (attribute, off by default) adds the reference's own rule behind these: when the identity key does not
match, the image is compared BY CONTENT with a private copy of the last image (``max|image - copy| == 0``, one device->host
synchronisation per frame exactly like mode_07.py:56-61) - for callers that re-upload an identical image every frame.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from .. import _capi
from .poser import PoseParameterGroup, Poser

OUT_CHANNELS = [4, 1, 4, 2, 4, 4, 4, 1, 4, 2, 4, 4, 1, 4, 4, 1, 4, 4, 2, 4, 1, 4, 4, 1, 4, 4, 2, 4, 1, 4, 4, 1, 4]
OUT_SIZE = [512] * 6 + [256] * 5 + [192] * 8 + [128] * 14


DEFAULT_EXACT_DECOMPOSER = "outer"


def _decomposer_mode(v):
    """Normalise the `exact_decomposer` argument: False / None / "" / "0" -> False, True / "all" / "1" -> "all", "outer" -> "outer"."""
    if v in (None, False, "", "0", "false", "off", "none", 0):
        return False
    if v in (True, "all", "1", "true", "on", 1):
        return "all"
    if v == "outer":
        return "outer"
    raise ValueError(f"exact_decomposer must be False, True / 'all' or 'outer', not {v!r}")


def default_exact_decomposer():
    """The library default of the mixed plan; `THA4_EXACT_DECOMPOSER=0|outer|all` in the environment overrides it (A/B measurements)."""
    v = os.environ.get("THA4_EXACT_DECOMPOSER")
    return DEFAULT_EXACT_DECOMPOSER if v is None or v == "" else _decomposer_mode(v)


class HipFullPoser(Poser):
    def __init__(self,
                 state_dict_loaders: Dict[str, Callable[[], Dict[str, np.ndarray]]],
                 device: torch.device,
                 pose_parameters: List[PoseParameterGroup],
                 eyebrow_morphed_image_index: int = 2,
                 default_output_index: int = 0,
                 max_batch: int = 1,
                 dtype: torch.dtype = torch.float,
                 exact_fp32: bool = False,
                 exact_decomposer: Optional[bool] = None):
        self.state_dict_loaders = state_dict_loaders
        #: THA4_FULL_EXACT_DECOMPOSER, the MIXED plan (round 6): only the eyebrow decomposer on the exact-fp32 kernels.  It carries ~90 % of the
        #: split plan's share of the posed frame's error (profiles/parity_r06/split_attribution*.txt) and its outputs are cached while the
        #: image is unchanged (mode_07.py:56-67): no cost per steady frame.  None = the library default (`default_exact_decomposer()`).
        #: True / "all": the whole decomposer; "outer": every convolution of it except the 16x16 bottleneck (THA4_FULL_EXACT_DECOMPOSER_OUTER); False: none.
        self.exact_decomposer = _decomposer_mode(default_exact_decomposer() if exact_decomposer is None else exact_decomposer)
        #: THA4_FULL_EXACT_FP32: every convolution on the exact-fp32 kernels (fp32's own operand range; ~2.5-3x slower).  The plan to
        #: fall back to when `check_numeric_range()` / `pose()` report THA4_ERR_NUMERIC_RANGE for weights that are fine in fp32:
        #: `poser.set_exact_fp32(True)` re-plans the handle on the next call.
        self.exact_fp32 = bool(exact_fp32)
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
        self._cache_image = None           # strong reference: keeps the keyed storage alive (mode_07.py:65 cached_batch_0)
        self._cache_copy = None            # content_cache: private copy of the image the cached decomposer outputs belong to
        self._cache_copy_batch = 0
        self.num_networks = 5
        self.first_output = 0              # C-ABI output index of list entry 0 (mode_12: 11)
        self.list_length = _capi.FULL_NUM_OUTPUTS

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

    def pose(self, image: Tensor, pose: Tensor, output_index: Optional[int] = None, image_changed: bool = False,
             image_version: Optional[int] = None) -> Tensor:
        if output_index is None:
            output_index = self.default_output_index
        return self._run(image, pose, [output_index], image_changed, image_version)[0]

    def get_posing_outputs(self, image: Tensor, pose: Tensor, image_changed: bool = False, indices=None,
                           image_version: Optional[int] = None) -> List[Tensor]:
        """The reference's list of outputs (mode_07.py:126-132).  ``indices`` (extension) restricts the work to those
        entries - only they are allocated and written - and returns them in the given order, e.g. ``(0, 1, 2, 3, 5)``
        for the distiller's teacher call (siren_morpher_protocols_03.py:56-72)."""
        wanted = list(range(self.list_length)) if indices is None else [int(i) for i in indices]
        return self._run(image, pose, wanted, image_changed, image_version)

    def pose_display_rgba8(self, image: Tensor, pose: Tensor, background_rgb=None, image_changed: bool = False,
                           image_version: Optional[int] = None) -> Tensor:
        """``pose()`` (output 0) followed by the puppeteers' display post-processing
        (character_model_ifacialmocap_puppeteer.py:325-349,377-381), fused into the upscaler's tail kernel
        (tha4_full_pose_ex / tha4_display): returns ``uint8 [B,512,512,4]``."""
        return self._run(image, pose, [], image_changed, image_version, display=(True, background_rgb))

    def check_numeric_range(self):
        """Synchronous form of the numeric-range guard (tha4_full_numeric_status): waits for the device and raises
        ``Tha4Error`` if a pose call since the last check left the fp16 hi/lo operand range (|normalised + activated value|
        >= 65520) or met NaN / inf.  ``pose()`` itself never synchronises: it raises for an EARLIER call's fault."""
        if self._handle is not None:
            _capi.check(self._lib, self._lib.tha4_full_numeric_status(self._handle, 1), "tha4_full_numeric_status")

    def set_timing(self, enable: bool = True):
        """Per-op HIP-event timing of the following pose calls (tha4_full_set_timing, ABI v5; a measurement aid: timed calls are slower).
        The handle must exist (pose once or `get_modules()` first); a re-created handle (regrow, `set_exact_fp32`) starts untimed."""
        if self._handle is None:
            raise _capi.Tha4Error("set_timing needs a live handle: call get_modules() or pose() first")
        _capi.check(self._lib, self._lib.tha4_full_set_timing(self._handle, int(bool(enable))), "tha4_full_set_timing")

    def op_info(self):
        """[(label, as-written GFLOP per frame)] for every op of the handle's schedule, decomposer ops first (tha4_full_op_info)."""
        if self._handle is None:
            raise _capi.Tha4Error("op_info needs a live handle")
        n = self._lib.tha4_full_num_ops(self._handle)
        out = []
        for i in range(n):
            label, gf = C.c_char_p(), C.c_double()
            _capi.check(self._lib, self._lib.tha4_full_op_info(self._handle, i, C.byref(label), C.byref(gf)), "tha4_full_op_info")
            out.append((label.value.decode(), float(gf.value)))
        return out

    def last_op_ms(self):
        """Milliseconds of every op of the LAST timed pose call (waits for it; 0 for decomposer ops the call reused): tha4_full_last_op_ms."""
        if self._handle is None:
            raise _capi.Tha4Error("last_op_ms needs a live handle")
        n = self._lib.tha4_full_num_ops(self._handle)
        buf = (C.c_float * n)()
        _capi.check(self._lib, self._lib.tha4_full_last_op_ms(self._handle, buf, n), "tha4_full_last_op_ms")
        return [float(x) for x in buf]

    def set_exact_fp32(self, on: bool = True) -> "HipFullPoser":
        """Switch between the default plan (fp16 hi/lo operand halves, |operand| <= 65504) and the exact-fp32 plan; the native handle
        is re-created lazily by the next call.  Results of the two plans agree within the parity gate, not bitwise."""
        if bool(on) != self.exact_fp32:
            self.exact_fp32 = bool(on)
            self._destroy_handle()
        return self

    def set_exact_decomposer(self, on: bool = True) -> "HipFullPoser":
        """Switch the mixed plan (eyebrow decomposer on the exact-fp32 kernels) on or off; the native handle is re-created lazily by the next call."""
        if _decomposer_mode(on) != self.exact_decomposer:
            self.exact_decomposer = _decomposer_mode(on)
            self._destroy_handle()
        return self

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
        self._cache_image = None
        self._cache_copy = None

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:
            pass

    #: what happens when a batch larger than ``max_batch`` arrives on a live handle.  The launch plan (K split, conv_small,
    #: folded normalisations) is chosen for the batch the handle is built for and a frame's bytes are only guaranteed
    #: within one plan, so growing is never silent: "warn" (default) re-creates the handle with a RuntimeWarning - the
    #: reference accepts any batch, a drop-in must too -, "error" raises Tha4Error instead (callers that rely on bitwise
    #: reproducibility, e.g. the sharded stream), "allow" re-creates quietly.
    regrow_policy = "warn"

    #: how a numeric fault of an earlier call is delivered (tha4_full_set_fault_policy): "refuse_next" (default) - the next pose()
    #: raises once and is not enqueued; "status_only" - pose() never refuses, `check_numeric_range()` is the only report (real-time
    #: callers that poll it and cannot lose a frame).  pose() never synchronises: its outputs are unchecked until the check is called.
    #: A property: the setter validates the value and, when the handle already exists, applies it to the LIVE handle at once
    #: (round-4 advisor finding: it used to be read only when a handle was created).
    _fault_policy = "refuse_next"

    @property
    def fault_policy(self) -> str:
        return self._fault_policy

    @fault_policy.setter
    def fault_policy(self, value: str):
        if value not in ("refuse_next", "status_only"):
            raise _capi.Tha4Error(f"unknown fault_policy {value!r} (expected 'refuse_next' or 'status_only')")
        self._fault_policy = value
        if getattr(self, "_handle", None) is not None:
            self._apply_fault_policy(self._handle)

    def _apply_fault_policy(self, handle):
        _capi.check(self._lib, self._lib.tha4_full_set_fault_policy(handle, 1 if self._fault_policy == "status_only" else 0),
                    "tha4_full_set_fault_policy")

    #: the reference's content rule for the eyebrow-decomposer cache (mode_07.py:56-61), behind the identity / version rules:
    #: costs one device->host synchronisation per call whose identity key misses, like the reference pays on every call
    content_cache = False

    def _ensure_handle(self, batch: int):
        if self._handle is not None and batch <= self._max_batch:
            return
        if self._handle is not None:
            msg = (f"HipFullPoser: batch {batch} exceeds the max_batch {self._max_batch} this handle was planned for; "
                   f"re-creating it for {batch} frames changes the launch plan - results stay within the parity gate but "
                   f"are no longer bitwise equal to frames posed before (pass max_batch= to create_poser to avoid this)")
            if self.regrow_policy == "error":
                raise _capi.Tha4Error(msg)
            if self.regrow_policy == "warn":
                import warnings
                warnings.warn(msg, RuntimeWarning, stacklevel=4)
        dev = self._device_index()
        if self._lib is None:
            self._lib = _capi.load_library()
        if self._state_dicts is None:
            self._state_dicts = {k: loader() for k, loader in self.state_dict_loaders.items()}
        self._destroy_handle()
        self._max_batch = max(self._max_batch, batch)
        weights, keep = _capi.build_full_weights(self._state_dicts)
        handle = C.c_void_p()
        st = self._lib.tha4_full_create_ex(C.byref(weights), self.eyebrow_morphed_image_index, dev, self._max_batch,
                                           self.num_networks, (_capi.FULL_EXACT_FP32 if self.exact_fp32 else 0) |
                                           {"all": _capi.FULL_EXACT_DECOMPOSER, "outer": _capi.FULL_EXACT_DECOMPOSER_OUTER, "": 0}[self.exact_decomposer or ""],
                                           C.byref(handle))
        _capi.check(self._lib, st, "tha4_full_create_ex")
        del keep
        self._handle = handle
        self._apply_fault_policy(handle)

    def _run(self, image: Tensor, pose: Tensor, wanted: List[int], image_changed: bool,
             image_version: Optional[int] = None, display=None):
        given = image                                       # the caller's tensor object (cache identity)
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
        dev = self._device_index()
        for name, t in (("image", image), ("pose", pose)):
            if t.dtype != torch.float32:
                raise AssertionError(f"{name} must be float32, got {t.dtype}")
            if t.device.type != "cuda" or t.device.index != dev:
                raise AssertionError(f"{name} is on {t.device}, poser is on {self.device}")
        if (not wanted and display is None) or any(i < 0 or i >= self.list_length for i in wanted):
            raise AssertionError(f"output indices must be in [0, {self.list_length}), got {wanted}")
        copied = not image.is_contiguous()
        image, pose = image.contiguous(), pose.contiguous()
        self._ensure_handle(b)
        # ---- eyebrow-decomposer cache (mode_07.py:56-67) ----
        if image_version is not None:
            key = ("version", int(image_version), tuple(image.shape), b)
            keep = None
        else:
            try:
                version = given._version
            except RuntimeError:                           # inference tensors do not track a version counter
                version = None
            key = None if (copied or version is None) else ("tensor", given.data_ptr(), tuple(given.shape), tuple(given.stride()), version, b)
            keep = given
        reuse = (not image_changed) and key is not None and key == self._cache_key
        if self.content_cache and not image_changed and not reuse and self._cache_copy is not None \
                and self._cache_copy_batch == b and self._cache_copy.shape == image.shape:
            reuse = bool((image - self._cache_copy).abs().max().item() == 0)      # the reference's test, with its sync
        # the private copy behind the content rule is replaced only AFTER the call below has succeeded: a failed call (e.g.
        # THA4_ERR_NUMERIC_RANGE returns without enqueueing anything) must not leave a copy that makes the NEXT call with the same
        # content claim a decomposer result that was never computed (round-3 advisor finding)
        new_copy = image.clone() if (self.content_cache and not reuse) else None      # private copy: immune to in-place edits
        target = torch.device("cuda", dev)
        outs = {}
        ptrs = (C.c_void_p * _capi.FULL_NUM_OUTPUTS)()
        for i in sorted(set(wanted)):
            ci = self.first_output + i                     # index in the C ABI's 33-entry list
            t = torch.empty((b, OUT_CHANNELS[ci], OUT_SIZE[ci], OUT_SIZE[ci]), dtype=torch.float32, device=target)
            outs[i] = t
            ptrs[ci] = t.data_ptr()
        stride = 0 if (image.shape[0] == 1 and b > 1) else 4 * 512 * 512
        rgba, disp_ref, disp_keep = None, None, None
        if display is not None:
            rgba = torch.empty((b, 512, 512, 4), dtype=torch.uint8, device=target)
            disp, disp_keep = _capi.make_display(rgba.data_ptr(), display[1])
            disp_ref = C.byref(disp)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            st = self._lib.tha4_full_pose_ex(self._handle, image.data_ptr(), stride, pose.data_ptr(), b, ptrs, int(reuse),
                                             disp_ref, C.c_void_p(stream))
        if st != 0:                                         # nothing of this call may be reused: forget every cache rule's state
            self._cache_key, self._cache_image, self._cache_copy = None, None, None
        _capi.check(self._lib, st, "tha4_full_pose_ex")
        del disp_keep
        if new_copy is not None:
            self._cache_copy, self._cache_copy_batch = new_copy, b
        self._cache_key = key
        self._cache_image = keep if key is not None else None
        if display is not None:
            return rgba
        return [outs[i] for i in wanted]
