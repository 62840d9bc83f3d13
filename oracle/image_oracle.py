"""CPU oracle for the image ingest / display epilogue rows.

*** TEST INFRASTRUCTURE ONLY ***  Parity status: PINNED BY LIVE REFERENCE.
  * ingest: tests/golden/student_lambda_0{0,1}_io.npz hold the decoded RGBA8 character image AND the tensor the
    reference's ``extract_pytorch_image_from_PIL_image`` produced from it;
  * display: tests/golden/display_io.npz holds what the reference's ``convert_linear_to_srgb`` /
    ``torch_linear_to_srgb`` (imported unmodified by tests/golden/make_golden_display.py) produce inside the
    puppeteers' post-processing sequence, for five backgrounds (tests/test_image_io.py).

Restated reference code (paths relative to /root/reference/src/tha4):
  shion/base/image_util.py:10-17,127-149,194-198   PIL RGBA8 -> fp32 poser input
  app/character_model_ifacialmocap_puppeteer.py:325-349,377-381 + image_util.py:56-58   display post-processing
"""
import numpy as np


def ingest_rgba8_numpy(rgba8: np.ndarray) -> np.ndarray:
    """[H,W,4] uint8 -> [4,H,W] float32, following extract_numpy_image_from_PIL_image op for op (fp32)."""
    img = (rgba8.astype(np.float32) / 255.0)
    x = np.clip(img[:, :, 0:3], 0.0, 1.0)
    img[:, :, 0:3] = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)
    img[:, :, 0:3] = img[:, :, 0:3] * img[:, :, 3:4]
    img = img * 2.0 + (-1.0)
    h, w, c = img.shape
    return img.reshape(h * w, c).transpose().reshape(c, h, w).astype(np.float32)


def display_rgba8_torch(frame, background_rgb=None):
    """[4,H,W] float32 torch tensor -> [H,W,4] uint8, the puppeteers' sequence of torch ops."""
    import torch
    x = torch.clip((frame.float() + 1.0) / 2.0, 0.0, 1.0)
    rgb = torch.clip(x[0:3], 0.0, 1.0)
    rgb = torch.where(torch.le(rgb, 0.003130804953560372), rgb * 12.92, 1.055 * (rgb ** (1.0 / 2.4)) - 0.055)
    x = torch.cat([rgb, x[3:4]], dim=0)
    if background_rgb is not None:
        bg = torch.tensor(background_rgb, dtype=torch.float32).view(3, 1, 1)
        alpha = x[3:4]
        x = torch.cat([x[0:3] * alpha + (1.0 - alpha) * bg, torch.ones_like(alpha)], dim=0)
    c, h, w = x.shape
    out = 255.0 * torch.transpose(x.reshape(c, h * w), 0, 1).reshape(h, w, c)
    return out.byte()
