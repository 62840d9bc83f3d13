"""ctypes binding of the C ABI declared in ``include/tha4_hip.h`` (libtha4_hip.so).

This is the only place the Python host side touches native code.  There is NO fallback:
if the shared library is missing or fails to load, importing a poser raises (the product path
must fail loudly rather than silently run something else).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtha4_hip.so")

THA4_ABI_VERSION = 6
STUDENT_EXACT_FP32 = 1
FULL_EXACT_FP32 = 1
FULL_EXACT_DECOMPOSER = 2
FULL_EXACT_DECOMPOSER_OUTER = 4

c_float_p = C.POINTER(C.c_float)


class Tha4Linear(C.Structure):
    _fields_ = [("weight", c_float_p), ("bias", c_float_p), ("out_ch", C.c_int32), ("in_ch", C.c_int32)]


class Tha4StudentWeights(C.Structure):
    _fields_ = [("face_sine", Tha4Linear * 8),
                ("face_last", Tha4Linear),
                ("body_sine", (Tha4Linear * 3) * 3),
                ("body_last", Tha4Linear)]


class Tha4PositionAxes(C.Structure):
    _fields_ = [("axis128", c_float_p), ("axis256", c_float_p), ("axis512", c_float_p)]


class Tha4Display(C.Structure):
    """tha4_display: the display epilogue fused into the kernel that composes the posed frame."""
    _fields_ = [("rgba8_dev", C.c_void_p), ("background_rgb", c_float_p)]


class Tha4StudentAux(C.Structure):
    _fields_ = [("alpha_dev", C.c_void_p), ("color_change_dev", C.c_void_p), ("warped_dev", C.c_void_p),
                ("grid_change_dev", C.c_void_p), ("face_dev", C.c_void_p), ("display", Tha4Display)]


ERR_NUMERIC_RANGE = -5


def make_display(rgba8_ptr: int, background_rgb):
    """(Tha4Display, keepalive) for a device uint8 [B,512,512,4] buffer and an optional 3-float background colour."""
    d = Tha4Display()
    d.rgba8_dev = rgba8_ptr
    keep = None
    if background_rgb is not None:
        keep = (C.c_float * 3)(*[float(x) for x in background_rgb])
        d.background_rgb = C.cast(keep, c_float_p)
    return d, keep


class Tha4NamedTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", c_float_p), ("ndim", C.c_int32), ("dims", C.c_int64 * 4)]


class Tha4FullWeights(C.Structure):
    _fields_ = [("tensors", C.POINTER(Tha4NamedTensor) * 5), ("counts", C.c_int32 * 5)]


FULL_NETWORKS = ["eyebrow_decomposer", "eyebrow_morphing_combiner", "face_morpher", "body_morpher", "upscaler"]
FULL_NUM_OUTPUTS = 33


class Tha4Error(RuntimeError):
    """Raised for every non-zero status coming back over the C ABI (the reference raises
    RuntimeError / AssertionError from Python for the same conditions)."""


def _linear(weight: np.ndarray, bias: np.ndarray, keep: list) -> Tha4Linear:
    w = np.ascontiguousarray(weight, dtype=np.float32)
    if w.ndim == 4:                      # Conv2d 1x1 kernel [O,I,1,1]
        w = w.reshape(w.shape[0], w.shape[1])
    b = np.ascontiguousarray(bias, dtype=np.float32)
    if w.ndim != 2 or b.ndim != 1 or b.shape[0] != w.shape[0]:
        raise Tha4Error(f"malformed linear layer: weight {weight.shape}, bias {bias.shape}")
    keep.extend([w, b])
    return Tha4Linear(w.ctypes.data_as(c_float_p), b.ctypes.data_as(c_float_p), w.shape[0], w.shape[1])


def build_student_weights(face_sd: Dict[str, np.ndarray], body_sd: Dict[str, np.ndarray]):
    """Map the two reference state_dicts (SURVEY.md Appendix B key layout) onto the C struct.
    Returns (struct, keepalive list).  Missing keys raise KeyError like load_state_dict would."""
    keep: list = []
    s = Tha4StudentWeights()
    for i in range(8):
        s.face_sine[i] = _linear(face_sd[f"siren.sine_layers.{i}.linear.weight"],
                                 face_sd[f"siren.sine_layers.{i}.linear.bias"], keep)
    s.face_last = _linear(face_sd["siren.last_linear.weight"], face_sd["siren.last_linear.bias"], keep)
    for l in range(3):
        for j in range(3):
            s.body_sine[l][j] = _linear(body_sd[f"siren_layers.{l}.{j}.linear.weight"],
                                        body_sd[f"siren_layers.{l}.{j}.linear.bias"], keep)
    s.body_last = _linear(body_sd["last_linear.weight"], body_sd["last_linear.bias"], keep)
    return s, keep


def build_full_weights(state_dicts: Dict[str, Dict[str, np.ndarray]]):
    """Map the five reference state_dicts (keys = mode_07.Network names, mode_07.py:24-29) onto the C struct."""
    keep: list = []
    s = Tha4FullWeights()
    for i, net in enumerate(FULL_NETWORKS):
        if net not in state_dicts:           # mode_12: the first three networks only
            continue
        sd = state_dicts[net]
        arr = (Tha4NamedTensor * len(sd))()
        for j, (k, v) in enumerate(sd.items()):
            a = np.ascontiguousarray(v, dtype=np.float32)
            if a.ndim < 1 or a.ndim > 4:
                raise Tha4Error(f"{net}.{k}: unsupported rank {a.ndim}")
            name = k.encode()
            keep.extend([a, name])
            arr[j].name = name
            arr[j].data = a.ctypes.data_as(c_float_p)
            arr[j].ndim = a.ndim
            for d in range(a.ndim):
                arr[j].dims[d] = a.shape[d]
        keep.append(arr)
        s.tensors[i] = C.cast(arr, C.POINTER(Tha4NamedTensor))
        s.counts[i] = len(sd)
    return s, keep


def build_position_axes(axes: Optional[Dict[int, np.ndarray]]):
    if not axes:
        return None, []
    keep = []
    s = Tha4PositionAxes()
    for size, field in ((128, "axis128"), (256, "axis256"), (512, "axis512")):
        a = axes.get(size)
        if a is None:
            continue
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.shape != (size,):
            raise Tha4Error(f"position axis {size} has shape {a.shape}")
        keep.append(a)
        setattr(s, field, a.ctypes.data_as(c_float_p))
    return s, keep


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libtha4_hip.so and declare every prototype of include/tha4_hip.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # libtha4_hip.so needs libamdhip64.so.7; PyTorch-ROCm bundles its own copy under the same soname.
    # Import torch FIRST so that one HIP runtime (torch's) serves both - streams, events and device
    # pointers are only interchangeable inside a single runtime instance.
    import torch  # noqa: F401
    p = path or os.environ.get("THA4_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise Tha4Error(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(p)
    lib.tha4_abi_version.restype = C.c_int
    lib.tha4_last_error.restype = C.c_char_p
    lib.tha4_student_create.restype = C.c_int
    lib.tha4_student_create.argtypes = [C.POINTER(Tha4StudentWeights), C.POINTER(Tha4PositionAxes), C.c_int, C.c_int,
                                        C.POINTER(C.c_void_p)]
    lib.tha4_student_create_ex.restype = C.c_int
    lib.tha4_student_create_ex.argtypes = [C.POINTER(Tha4StudentWeights), C.POINTER(Tha4PositionAxes), C.c_int, C.c_int, C.c_int,
                                           C.POINTER(C.c_void_p)]
    lib.tha4_student_pose.restype = C.c_int
    lib.tha4_student_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                      C.POINTER(Tha4StudentAux), C.c_void_p]
    lib.tha4_student_set_weights.restype = C.c_int
    lib.tha4_student_set_weights.argtypes = [C.c_void_p, C.POINTER(Tha4StudentWeights)]
    lib.tha4_student_destroy.restype = None
    lib.tha4_student_destroy.argtypes = [C.c_void_p]
    lib.tha4_student_hand_off_scale.restype = C.c_float
    lib.tha4_student_hand_off_scale.argtypes = [C.c_void_p]
    lib.tha4_student_debug_read.restype = C.c_int
    lib.tha4_student_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.tha4_student_max_batch.restype = C.c_int
    lib.tha4_student_max_batch.argtypes = [C.c_void_p]
    lib.tha4_student_device.restype = C.c_int
    lib.tha4_student_device.argtypes = [C.c_void_p]
    lib.tha4_student_set_timing.restype = C.c_int
    lib.tha4_student_set_timing.argtypes = [C.c_void_p, C.c_int]
    lib.tha4_student_last_ms.restype = C.c_int
    lib.tha4_student_last_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    lib.tha4_full_create.restype = C.c_int
    lib.tha4_full_create.argtypes = [C.POINTER(Tha4FullWeights), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.tha4_full_create_ex.restype = C.c_int
    lib.tha4_full_create_ex.argtypes = [C.POINTER(Tha4FullWeights), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    lib.tha4_full_flags.restype = C.c_int
    lib.tha4_full_flags.argtypes = [C.c_void_p]
    lib.tha4_full_set_fault_policy.restype = C.c_int
    lib.tha4_full_set_fault_policy.argtypes = [C.c_void_p, C.c_int]
    lib.tha4_full_set_timing.restype = C.c_int
    lib.tha4_full_set_timing.argtypes = [C.c_void_p, C.c_int]
    lib.tha4_full_num_ops.restype = C.c_int
    lib.tha4_full_num_ops.argtypes = [C.c_void_p]
    lib.tha4_full_op_info.restype = C.c_int
    lib.tha4_full_op_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double)]
    lib.tha4_full_last_op_ms.restype = C.c_int
    lib.tha4_full_last_op_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
    lib.tha4_full_num_networks.restype = C.c_int
    lib.tha4_full_num_networks.argtypes = [C.c_void_p]
    lib.tha4_full_pose.restype = C.c_int
    lib.tha4_full_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int,
                                   C.c_void_p]
    lib.tha4_full_pose_ex.restype = C.c_int
    lib.tha4_full_pose_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int,
                                      C.POINTER(Tha4Display), C.c_void_p]
    lib.tha4_full_numeric_status.restype = C.c_int
    lib.tha4_full_numeric_status.argtypes = [C.c_void_p, C.c_int]
    lib.tha4_full_destroy.restype = None
    lib.tha4_full_destroy.argtypes = [C.c_void_p]
    lib.tha4_full_max_batch.restype = C.c_int
    lib.tha4_full_max_batch.argtypes = [C.c_void_p]
    lib.tha4_display_rgba8.restype = C.c_int
    lib.tha4_display_rgba8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, c_float_p, C.c_void_p, C.c_void_p]
    lib.tha4_ingest_rgba8.restype = C.c_int
    lib.tha4_ingest_rgba8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    v = lib.tha4_abi_version()
    if v != THA4_ABI_VERSION:
        raise Tha4Error(f"libtha4_hip.so ABI version {v} != expected {THA4_ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "tha4_abi_version", "tha4_last_error", "tha4_student_create", "tha4_student_create_ex", "tha4_student_pose", "tha4_student_set_weights", "tha4_student_destroy",
    "tha4_student_max_batch", "tha4_student_device", "tha4_student_debug_read", "tha4_student_hand_off_scale", "tha4_student_set_timing", "tha4_student_last_ms",
    "tha4_full_create", "tha4_full_create_ex", "tha4_full_num_networks", "tha4_full_flags", "tha4_full_set_fault_policy", "tha4_full_set_timing", "tha4_full_num_ops", "tha4_full_op_info", "tha4_full_last_op_ms", "tha4_full_pose", "tha4_full_pose_ex", "tha4_full_numeric_status",
    "tha4_full_destroy", "tha4_full_max_batch",
    "tha4_display_rgba8", "tha4_ingest_rgba8",
]


def check(lib: C.CDLL, status: int, what: str) -> None:
    if status != 0:
        msg = lib.tha4_last_error()
        raise Tha4Error(f"{what} failed (status {status}): {msg.decode() if msg else ''}")
