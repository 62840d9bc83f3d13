"""Image ingest and display epilogue around the poser path, on the GPU.

``to_display_rgba8``  mirrors the post-processing of the reference puppeteers
(src/tha4/app/character_model_ifacialmocap_puppeteer.py:325-349,377-381): posed frame ->
clip((x+1)/2) -> linear->sRGB -> optional opaque background -> HWC uint8.  Fused into one pass it
turns the 4 MiB fp32 frame into the 1 MiB the GUI / encoder / gather actually needs.
``image_from_rgba8`` / ``image_from_pil`` mirror ``extract_pytorch_image_from_PIL_image``
(src/tha4/shion/base/image_util.py:127-149,194-198).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from . import _capi


def _stream(t: Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def to_display_rgba8(frames: Tensor, background_rgb: Optional[Sequence[float]] = None) -> Tensor:
    """frames fp32 [B,4,H,W] or [4,H,W] on a ROCm device -> uint8 [B,H,W,4] (or [H,W,4])."""
    lib = _capi.load_library()
    squeeze = frames.dim() == 3
    if squeeze:
        frames = frames.unsqueeze(0)
    if frames.dim() != 4 or frames.shape[1] != 4 or frames.dtype != torch.float32 or not frames.is_cuda:
        raise AssertionError("frames must be a float32 [B,4,H,W] tensor on the GPU")
    frames = frames.contiguous()
    b, _, h, w = frames.shape
    out = torch.empty((b, h, w, 4), dtype=torch.uint8, device=frames.device)
    bg = None
    if background_rgb is not None:
        bg = (C.c_float * 3)(*[float(v) for v in background_rgb])
    _capi.check(lib, lib.tha4_display_rgba8(frames.data_ptr(), b, h, w, bg, out.data_ptr(), _stream(frames)), "tha4_display_rgba8")
    return out[0] if squeeze else out


def image_from_rgba8(rgba: Tensor) -> Tensor:
    """uint8 [B,H,W,4] or [H,W,4] on the GPU -> fp32 [B,4,H,W] (or [4,H,W]) in the poser's input convention
    (linear RGB premultiplied by alpha, scaled to [-1,1])."""
    lib = _capi.load_library()
    squeeze = rgba.dim() == 3
    if squeeze:
        rgba = rgba.unsqueeze(0)
    if rgba.dim() != 4 or rgba.shape[3] != 4 or rgba.dtype != torch.uint8 or not rgba.is_cuda:
        raise AssertionError("rgba must be a uint8 [B,H,W,4] tensor on the GPU")
    rgba = rgba.contiguous()
    b, h, w, _ = rgba.shape
    out = torch.empty((b, 4, h, w), dtype=torch.float32, device=rgba.device)
    _capi.check(lib, lib.tha4_ingest_rgba8(rgba.data_ptr(), b, h, w, out.data_ptr(), _stream(rgba)), "tha4_ingest_rgba8")
    return out[0] if squeeze else out


def resize_PIL_image(pil_image, size=(512, 512)):
    """Centre-crop to a square and Lanczos-resize, as the reference does for character images that are not already
    512x512 (src/tha4/image_util.py:29-33; its default size is 256, the posers need 512).  Host-side PIL call: the
    resampling kernel is PIL's, exactly as in the reference."""
    import PIL.Image
    w, h = pil_image.size
    d = min(w, h)
    r = ((w - d) // 2, (h - d) // 2, (w + d) // 2, (h + d) // 2)
    return pil_image.resize(size, resample=PIL.Image.LANCZOS, box=r)


def image_from_pil(pil_image, device: torch.device) -> Tensor:
    """PIL RGBA image -> poser input tensor [4,H,W] on `device` (the reference raises for non-RGBA character images,
    charmodel/character_model.py:38-39)."""
    if pil_image.mode != "RGBA":
        raise RuntimeError("Character image is not an RGBA image!")
    arr = np.asarray(pil_image, dtype=np.uint8)
    return image_from_rgba8(torch.from_numpy(np.ascontiguousarray(arr)).to(device))
